#!/usr/bin/env python3
"""`groth16.Prove` alone at 2^log_n wires / constraints (bench.py's synthetic key: 256 distinct points tiled, random operands):
    python tools/groth16_quickbench.py [log_n=22] [reps=4]
runs zklc_amd.groth16.Groth16Prover in the fixed-base form (default) and the plain form (ZKLC_GROTH16_FIXED=0) on the same key and
operands, prints the stage times of each and checks that the two proofs are the same eight words (size-independent parity: the
affine result of a multi-exponentiation is canonical).  Under rocprofv3 --kernel-trace --stats it gives the per-kernel split."""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import zklc_amd
from oracle import bn254 as B          # test data only: 256 points of each group
from zklc_amd.groth16 import Groth16Prover

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1 << lg
ctx = zklc_amd.Context(0)
rng = np.random.default_rng(3)
cur, step, g2 = B.g2_mul(12345, B.G2), B.g2_mul(777, B.G2), []
for _ in range(256):
    g2.append(B.g2_to_words(cur))
    cur = B.g2_add(cur, step)
cur, step, g1 = B.mul(54321, B.G1), B.mul(999, B.G1), []
for _ in range(256):
    g1.append(B.to_mont_words(cur[0]) + B.to_mont_words(cur[1]))
    cur = B.add(cur, step)
g1p, g2p = np.array(g1, dtype=np.uint64), np.array(g2, dtype=np.uint64)
tile1 = lambda k: np.tile(g1p, ((k + 255) // 256, 1))[:k]
pk = {"n": n, "n_public": 4, "A_words": tile1(n), "B1_words": tile1(n), "K_words": tile1(n - 5), "Z_words": tile1(n - 1),
      "B2_words": np.tile(g2p, ((n + 255) // 256, 1))[:n], "alpha1_words": g1p[1:2], "beta1_words": g1p[2:3], "delta1_words": g1p[3:4],
      "beta2_words": g2p[1:2], "delta2_words": g2p[2:3]}
w = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
w[:, 3] &= np.uint64((1 << 60) - 1)
w[0] = [1, 0, 0, 0]
abc = tuple(rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) & np.uint64((1 << 61) - 1) for _ in range(3))
proofs = {}
for mode in [m for m in os.environ.get("MODES", "1,0").split(",") if m]:
    os.environ["ZKLC_GROTH16_FIXED"] = mode
    t0 = time.perf_counter()
    gp = Groth16Prover(ctx, pk)
    t_key = time.perf_counter() - t0
    gp.prove_words(w, abc, 12345, 67890)
    best, stages = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        proofs[mode] = gp.prove_words(w, abc, 12345, 67890)
        dt = (time.perf_counter() - t0) * 1e3
        if best is None or dt < best:
            best, stages = dt, dict(gp.last_ms)
    print("groth16 prove 2^%d  fixed_base=%s: best of %d = %.2f ms   key resident after %.2f s   stages %s"
          % (lg, mode, reps, best, t_key, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in stages.items()}), flush=True)
    gp.close()
    del gp
if len(proofs) == 2:
    assert proofs["1"] == proofs["0"], "fixed-base proof differs from the plain form"
    print("fixed-base proof == plain-form proof (8 words)")
