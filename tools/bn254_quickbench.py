#!/usr/bin/env python3
"""G2 MSM, Fr NTT and pairing-check timings (row a10):  python tools/bn254_quickbench.py [g2_log=18] [ntt_log=22] [checks=4096]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
import zklc_amd
from oracle import bn254 as B

g2_log = int(sys.argv[1]) if len(sys.argv) > 1 else 18
ntt_log = int(sys.argv[2]) if len(sys.argv) > 2 else 22
checks = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
ctx = zklc_amd.Context(0)
dev = torch.device("cuda", 0)
lib = zklc_amd.load()
sp = ctx.stream_ptr()


def timed(fn, reps=3):
    fn()
    ctx.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


# ---- G2 MSM: 2^g2_log points = a few hundred distinct points tiled (generation in Python is slow), uniform scalars
base_n = 256
cur, step, pts = B.g2_mul(12345, B.G2), B.g2_mul(777, B.G2), []
for _ in range(base_n):
    pts.append(B.g2_to_words(cur))
    cur = B.g2_add(cur, step)
n = 1 << g2_log
pa = np.tile(np.array(pts, dtype=np.uint64), (n // base_n, 1))
rng = np.random.default_rng(1)
sc = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 60) - 1)
d_p = torch.from_numpy(pa.view(np.int64)).to(dev)
d_s = torch.from_numpy(sc.view(np.int64)).to(dev)
wb = int(lib.zklc_bn254_g2_msm_workspace_bytes(n))
d_w = torch.empty(wb, dtype=torch.uint8, device=dev)
d_o = torch.zeros(17, dtype=torch.int64, device=dev)
ms = timed(lambda: ctx._check(lib.zklc_bn254_g2_msm_dev(ctx._h, sp, d_p.data_ptr(), d_s.data_ptr(), n, d_o.data_ptr(),
                                                        d_o.data_ptr() + 128, d_w.data_ptr(), wb)))
print("G2 MSM 2^%d: %.2f ms  %.2f Melem/s  (160 B/element algorithmic = %.1f GB/s)" % (g2_log, ms, n / ms / 1e3, 160 * n / ms / 1e6))
del d_p, d_s, d_w

# ---- Fr NTT
n = 1 << ntt_log
a = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
a[:, 3] &= np.uint64((1 << 60) - 1)
d_a = torch.from_numpy(a.view(np.int64)).to(dev)
wb = int(lib.zklc_bn254_fr_ntt_workspace_bytes(ntt_log))
d_w = torch.empty(wb, dtype=torch.uint8, device=dev)
ms = timed(lambda: ctx._check(lib.zklc_bn254_fr_ntt_dev(ctx._h, sp, d_a.data_ptr(), ntt_log, 0, 1, d_w.data_ptr(), wb)))
print("Fr coset NTT 2^%d: %.2f ms  %.1f GB/s (64 B/element algorithmic)  %.2f Gbutterfly/s" %
      (ntt_log, ms, 64 * n / ms / 1e6, n // 2 * ntt_log / ms / 1e6))
del d_a, d_w

# ---- pairing checks (k = 4, the Groth16 shape)
p, q = B.mul(0xABCDEF, B.G1), B.g2_mul(0x13579B, B.G2)
one = [(p, q), (B.neg(p), q), (B.mul(5, B.G1), B.G2), (B.neg(B.G1), B.g2_mul(5, B.G2))]
g1 = np.array([[B.to_mont_words(x[0]) + B.to_mont_words(x[1]) for x, _ in one]] * checks, dtype=np.uint64)
g2 = np.array([[B.g2_to_words(y) for _, y in one]] * checks, dtype=np.uint64)
d1 = torch.from_numpy(g1.view(np.int64)).to(dev)
d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
d_r = torch.zeros(checks, dtype=torch.int32, device=dev)
ms = timed(lambda: ctx._check(lib.zklc_bn254_pairing_check_dev(ctx._h, sp, d1.data_ptr(), d2.data_ptr(), 4, checks, d_r.data_ptr(), None)), reps=2)
assert int(d_r.sum()) == checks
print("pairing checks (4 pairings each) x%d: %.1f ms  %.0f checks/s  (one check alone: latency-bound)" % (checks, ms, checks / ms * 1e3))
ms1 = timed(lambda: ctx._check(lib.zklc_bn254_pairing_check_dev(ctx._h, sp, d1.data_ptr(), d2.data_ptr(), 4, 1, d_r.data_ptr(), None)), reps=2)
print("single check latency: %.1f ms" % ms1)
