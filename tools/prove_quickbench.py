#!/usr/bin/env python3
"""Times zklc_plonky2_prove on synthetic circuits of the reference's two shapes (see zklc_amd/plonky2/synthetic.py):
   python tools/prove_quickbench.py [ed_bits=17] [rec_bits=12] [reps=5]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import zklc_amd
from zklc_amd.plonky2 import synthetic as SY, standard_recursion_config, wide_ecc_config, HASH_GL, HASH_BN128
from oracle import plonky2_verifier as V

ed_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 17
rec_bits = int(sys.argv[2]) if len(sys.argv) > 2 else 12
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = zklc_amd.Context(0)
for name, bits, cfg, mixf, hasher, npi in [("recursion 2^%d x 135 (Poseidon-GL)" % rec_bits, rec_bits, standard_recursion_config(), SY.recursion_shape_mix, HASH_GL, 16),
                                          ("recursion 2^%d x 135 (Poseidon-BN128 wrap)" % rec_bits, rec_bits, standard_recursion_config(), SY.recursion_shape_mix, HASH_BN128, 16),
                                          ("ed25519 2^%d x 234 (Poseidon-GL)" % ed_bits, ed_bits, wide_ecc_config(), SY.ed25519_shape_mix, HASH_GL, 584)]:
    t = time.time()
    data, wires, pis = SY.synthetic_circuit(bits, cfg, mixf(cfg), num_public_inputs=npi, seed=1)
    t_build = time.time() - t
    t = time.time()
    prover = data.prover(ctx, hasher)
    t_pre = time.time() - t
    best, tm = None, None
    for r in range(reps):
        t = time.time()
        pb = prover.prove_bytes(wires, pis)
        dt = time.time() - t
        if best is None or dt < best:
            best, tm = dt, prover.last_timings()
    t = time.time()
    V.verify(json.loads(json.dumps(zklc_amd.plonky2.serialization.proof_from_bytes(pb, prover.common, hasher))), prover.verifier_data(), prover.common)
    t_ver = time.time() - t
    print("%s: build %.1fs  preprocess %.3fs  prove %.2f ms (best of %d, host->device witness copy included)  proof %d B  oracle verify %.1fs OK"
          % (name, t_build, t_pre, best * 1e3, reps, len(pb), t_ver))
    print("   stages ms:", {k: round(v, 3) for k, v in tm.items()})
    prover.close()
