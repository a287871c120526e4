#!/usr/bin/env python3
"""the reference's per-signature Ed25519 circuit (2^18 x 234, 20 gate types), a few proofs of one real signature -- run under
   rocprofv3 --kernel-trace --stats to get the per-kernel split:   python tools/prove_profile_ed25519.py [reps]"""
import json
import os
import sys
sys.path.insert(0, ".")
import zklc_amd
from zklc_amd.plonky2 import CircuitBuilder, HASH_GL, wide_ecc_config, ed25519_circuit as E
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
j = json.load(open(os.path.join("tests", "golden", "ed25519_near_c1_small.json")))
msg, e = bytes.fromhex(j["msg"]), j["entries"][0]
b = CircuitBuilder(wide_ecc_config())
targets = E.ed25519_circuit(b, 8 * len(msg))
data = b.build()
fill = E.fill_ecdsa_targets(targets, msg, bytes.fromhex(e["approval"])[2:], bytes.fromhex(e["validator_tail"])[1:33])
data.witness_program(fill)
wires, pis = data.generate_witness_native([fill])
ctx = zklc_amd.Context(0)
prover = data.prover(ctx, HASH_GL)
for _ in range(reps):
    prover.prove_bytes(wires[0], [int(x) for x in pis[0]])
print("gates:", [g.id()[:40] for g in data.gates])
print(prover.last_timings())
