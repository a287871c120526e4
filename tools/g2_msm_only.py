#!/usr/bin/env python3
"""G2 multi-exponentiations alone (for rocprofv3 --kernel-trace --stats / --pmc passes):  python tools/g2_msm_only.py [log_n=18] [reps=4]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
import zklc_amd
from oracle import bn254 as B

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = zklc_amd.Context(0)
dev = torch.device("cuda", 0)
lib = zklc_amd.load()
sp = ctx.stream_ptr()
cur, step, pts = B.g2_mul(12345, B.G2), B.g2_mul(777, B.G2), []
for _ in range(256):
    pts.append(B.g2_to_words(cur))
    cur = B.g2_add(cur, step)
n = 1 << lg
pa = np.tile(np.array(pts, dtype=np.uint64), (max(1, n // 256), 1))[:n]
rng = np.random.default_rng(1)
sc = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
sc[:, 3] &= np.uint64((1 << 60) - 1)
d_p = torch.from_numpy(pa.view(np.int64)).to(dev)
d_s = torch.from_numpy(sc.view(np.int64)).to(dev)
wb = int(lib.zklc_bn254_g2_msm_workspace_bytes(n))
d_w = torch.empty(wb, dtype=torch.uint8, device=dev)
d_o = torch.zeros(17, dtype=torch.int64, device=dev)
fn = lambda: ctx._check(lib.zklc_bn254_g2_msm_dev(ctx._h, sp, d_p.data_ptr(), d_s.data_ptr(), n, d_o.data_ptr(), d_o.data_ptr() + 128,
                                                  d_w.data_ptr(), wb))
fn()
ctx.synchronize()
t = time.perf_counter()
for _ in range(reps - 1):
    fn()
ctx.synchronize()
ms = (time.perf_counter() - t) / max(1, reps - 1) * 1e3
print("G2 MSM 2^%d: %.2f ms  %.2f Melem/s  (%d launches of the pipeline)" % (lg, ms, n / ms / 1e3, reps))
if lg <= 12:        # parity against the oracle's Python G2 arithmetic (small sizes only)
    vals = [sum(int(sc[i, k]) << (64 * k) for k in range(4)) for i in range(n)]
    plist = [B.g2_from_words([int(x) for x in pa[i]]) for i in range(n)]
    want = B.g2_to_words(B.g2_msm(vals, plist))
    assert [int(x) for x in d_o[:16].cpu().numpy().view(np.uint64)] == [int(x) for x in want]
    print("   equals the oracle")
