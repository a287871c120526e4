"""Quick timing of the BN254 G1 MSM at C4 sizes (not the driver's bench)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zklc_amd  # noqa: E402
from oracle import cport  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
dists = next((a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--dist=")), ["U"])
variants = next((a.split("=", 1)[1].split(",") for a in sys.argv[1:] if a.startswith("--variants=")), [None])
logs = [int(x) for x in args] or [16, 18, 20, 22]
nmax = 1 << max(logs)
t0 = time.time()
pts_h = cport.bn254_gen_points(nmax, 5, 3)
print("generated %d points in %.1f s" % (nmax, time.time() - t0), flush=True)
rng = np.random.default_rng(1)
sc_h = rng.integers(0, 2**63, size=(nmax, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(nmax, 4), dtype=np.uint64)
sc_h[:, 3] &= np.uint64((1 << 60) - 1)


def scalars_of(dist):
    """(U) uniform, (W) witness-like: 50 % in {0, 1}, 30 % < 2^64, 20 % uniform, (A1) all equal, (A2) < 2^64"""
    s = sc_h.copy()
    if dist == "W":
        kind = rng.random(nmax)
        small = kind < 0.5
        s[small] = 0
        s[small, 0] = rng.integers(0, 2, size=int(small.sum()), dtype=np.uint64)
        s[(kind >= 0.5) & (kind < 0.8), 1:] = 0
    elif dist == "A1":
        s[:] = s[0]
    elif dist == "A2":
        s[:, 1:] = 0
    return s


pts = torch.from_numpy(pts_h.view(np.int64)).cuda()
with zklc_amd.Context(0) as c:
    st = torch.cuda.Stream()
    for lg, variant, dist in [(lg, v, d) for lg in logs for v in variants for d in dists]:
        sc_d = scalars_of(dist)
        sc = torch.from_numpy(sc_d.view(np.int64)).cuda()
        if variant is not None:
            os.environ["ZKLC_MSM_WAVES"] = variant       # A/B switch of the slice kernel (waves per SIMD), read by the library at every call
        n = 1 << lg
        wb = c.bn254_g1_msm_workspace_bytes(n)
        ws = torch.empty(wb, dtype=torch.uint8, device="cuda")
        out = torch.zeros(8, dtype=torch.int64, device="cuda")
        inf = torch.zeros(1, dtype=torch.int32, device="cuda")
        fn = lambda: c.bn254_g1_msm_dev(pts, sc, n, out, inf, ws, wb, stream=st)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 3
        e0.record(st)
        for _ in range(iters):
            fn()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print("MSM 2^%d (%s)%s: %.2f ms  %.2f Melem/s  workspace %.0f MB" % (lg, dist, "" if variant is None else " [slice kernel, %s waves per SIMD]" % variant,
                                                                         ms, n / ms / 1e3, wb / 1e6), flush=True)
        if "--fixed" in sys.argv:
            # the fixed-base form (round 5): table of 2^(c w) P_i built once, one bucket set for all windows
            plain = out.cpu().numpy().view(np.uint64).copy()
            t0 = time.time()
            table = c.bn254_msm_fixed_table(pts[:n], n, stream=st)
            torch.cuda.synchronize()
            t_tab = time.time() - t0
            ffn = lambda: c.bn254_msm_fixed_dev(table, sc, n, out, inf, ws, wb, stream=st)
            ffn()
            torch.cuda.synchronize()
            e0.record(st)
            for _ in range(iters):
                ffn()
            e1.record(st)
            torch.cuda.synchronize()
            fms = e0.elapsed_time(e1) / iters
            same = bool(np.array_equal(out.cpu().numpy().view(np.uint64), plain))
            print("   fixed-base: %.2f ms  %.2f Melem/s  (table %.0f MB built in %.2f s)  equals the plain form: %s" % (
                fms, n / fms / 1e3, table.numel() / 1e6, t_tab, same), flush=True)
            del table
        if lg <= 20:
            t0 = time.time()
            want, winf, used = cport.bn254_msm(pts_h[:n], sc_d[:n], nthreads=16)
            dt = time.time() - t0
            got = out.cpu().numpy().view(np.uint64)
            print("   oracle (C, %d threads): %.2f s  %.3f Melem/s  match=%s" % (used, dt, n / dt / 1e6, bool(np.array_equal(got, want))), flush=True)
