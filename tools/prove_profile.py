#!/usr/bin/env python3
"""one circuit, a few proofs -- run under rocprofv3 --kernel-trace --stats to get the per-kernel split of a proof
   python tools/prove_profile.py [ed|rec|wrap] [bits] [reps]"""
import sys
sys.path.insert(0, ".")
import zklc_amd
from zklc_amd.plonky2 import synthetic as SY, standard_recursion_config, wide_ecc_config, HASH_GL, HASH_BN128
which = sys.argv[1] if len(sys.argv) > 1 else "ed"
bits = int(sys.argv[2]) if len(sys.argv) > 2 else (17 if which == "ed" else 12)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = zklc_amd.Context(0)
if which == "ed":
    cfg, mix, hasher, npi = wide_ecc_config(), SY.ed25519_shape_mix, HASH_GL, 584
else:
    cfg, mix, hasher, npi = standard_recursion_config(), SY.recursion_shape_mix, (HASH_BN128 if which == "wrap" else HASH_GL), 16
data, wires, pis = SY.synthetic_circuit(bits, cfg, mix(cfg), num_public_inputs=npi, seed=1)
prover = data.prover(ctx, hasher)
for _ in range(reps):
    prover.prove_bytes(wires, pis)
print(which, bits, prover.last_timings())
