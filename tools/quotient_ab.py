#!/usr/bin/env python3
"""A/B of the quotient phase on the reference's Ed25519 circuit (2^18 x 234, 20 gate types, real signature): per-gate launches vs
the fused LDS-tile kernel with 4 / 8 waves per workgroup.  The proof bytes must be identical.  python tools/quotient_ab.py [reps]"""
import hashlib
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
os.environ["ZKLC_P2_DEBUG"] = "1"
ref = None
for mode, waves in (("pergate", None), ("fused", "4"), ("fused", "8")):  # default = pergate
    os.environ["ZKLC_P2_QUOTIENT"] = mode
    if waves:
        os.environ["ZKLC_P2_FQ_WAVES"] = waves
    prover = data.prover(ctx, HASH_GL)
    best = None
    for _ in range(reps):
        raw = prover.prove_bytes(wires[0], [int(x) for x in pis[0]])
        t = prover.last_timings()
        if best is None or t["total"] < best["total"]:
            best = t
    h = hashlib.sha256(raw).hexdigest()[:16]
    ref = ref or h
    print("%-8s waves=%s  proof sha256 %s %s  stages ms %s" % (mode, waves, h, "== per-gate" if h == ref else "DIFFERS", {k: round(v, 2) for k, v in best.items()}), flush=True)
    assert h == ref
    prover.close()
