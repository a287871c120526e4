#!/usr/bin/env python3
"""One mode of the U32AddMany quotient evaluator on the reference's Ed25519 circuit (2^18 x 234, eight AddMany variants, a real NEAR
signature): prints the proof's sha256 and the best stage times.  The mode is read by the library once per process:

    ZKLC_P2_ADDMANY=pergate python tools/addmany_ab.py [reps]      # one launch per variant list (rounds 1-3)
    python tools/addmany_ab.py [reps]                              # the LDS-tile kernel (default since round 4)
The hashes of the two runs must be equal (and equal the C prover's: tests/test_gpu_plonky2.py)."""
import hashlib
import json
import os
import sys
sys.path.insert(0, ".")
os.environ.setdefault("ZKLC_CIRCUIT_CACHE", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".circuit_cache"))
import zklc_amd
from zklc_amd.signatures import ApprovalProver
from zklc_amd.plonky2 import ed25519_circuit as E
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
j = json.load(open(os.path.join("tests", "golden", "ed25519_near_c1_small.json")))
msg, e = bytes.fromhex(j["msg"]), j["entries"][0]
ctx = zklc_amd.Context(0)
ap = ApprovalProver(ctx)
data, targets, prover, _ = ap.ed25519_circuit(len(msg))
fill = E.fill_ecdsa_targets(targets, msg, bytes.fromhex(e["approval"])[2:], bytes.fromhex(e["validator_tail"])[1:33])
wires, pis = data.generate_witness_native([fill])
best = None
for _ in range(reps):
    raw = prover.prove_bytes(wires[0], [int(x) for x in pis[0]])
    t = prover.last_timings()
    if best is None or t["total"] < best["total"]:
        best = t
print("ZKLC_P2_ADDMANY=%s  proof sha256 %s  stages ms %s" % (os.environ.get("ZKLC_P2_ADDMANY", "(default: tile)"), hashlib.sha256(raw).hexdigest()[:16],
                                                              {k: round(v, 2) for k, v in best.items()}), flush=True)
ap.close()
