#!/usr/bin/env python3
"""Evaluation points that pin the reference's FIXED wrap circuit (BUILD container only; reads /root/reference).

All three golden final proofs of the reference -- near_bft_finality/proofs/{random/CGZP..,epoch/4RjX..,epoch/CbAH..} -- share ONE
verifier_data.json: the last recursion (bin/prove_block.rs:279-287 -> prove_crypto/recursion.rs:36-94) is a fixed circuit.  Beside its
Merkle cap and digest, every proof reveals values of the circuit's 5 constant / selector and 80 sigma polynomials:
  * openings.constants / openings.plonk_sigmas at the proof's challenge point zeta (extension field), and
  * the leaves of the constants_sigmas tree at its 28 FRI query positions (base-field points g * w^bitrev(index) of the 2^15 coset).
3 x (1 + 28) = 87 points per polynomial: a candidate circuit instance (row -> gate map, gate constants, copy classes) can be checked
COLUMN BY COLUMN against them (tools/wrap_instance.py), long before its Merkle cap can match.  Output:
tests/golden/plonky2_wrap_instance_points.json."""
import json
import os
import sys

REF = "/root/reference/near_bft_finality/proofs"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
from oracle import goldilocks as gl  # noqa: E402
from oracle import plonky2_verifier as V  # noqa: E402

DIRS = ["random/CGZPhFRkL3NvmGaXWBc6N7qJD519EUe6vyNpaEyDe2Ev", "epoch/4RjXBrNcu39wutFTuFpnRHgNqgHxLMcGBKNEQdtkSBhy",
        "epoch/CbAHBGJ8VQot2m6KhH9PLasMgcDtkPJBfp9bjAEMJ8UK"]


def main():
    points, vds = [], []
    for d in DIRS:
        p = json.load(open(os.path.join(REF, d, "proof.json")))
        v = json.load(open(os.path.join(REF, d, "verifier_data.json")))
        c = json.load(open(os.path.join(REF, d, "common_data.json")))
        vds.append(v)
        pf = V.parse_proof(p, v)
        ch = V.challenges(pf, c)
        o = pf["openings"]
        points.append({"proof": d, "kind": "zeta", "x": list(ch["zeta"]), "values": [list(e) for e in o["constants"] + o["plonk_sigmas"]]})
        n_log = c["fri_params"]["degree_bits"] + c["fri_params"]["config"]["rate_bits"]
        for rnd, (init, _) in enumerate(pf["rounds"]):
            x_index = ch["query_indices"][rnd] % (1 << n_log)
            rev = int(format(x_index, "0%db" % n_log)[::-1], 2)
            x = gl.GENERATOR * pow(gl.root_of_unity(n_log), rev, gl.P) % gl.P
            points.append({"proof": d, "kind": "fri_query", "x": [x, 0], "values": [[int(e), 0] for e in init[0][0]]})
    assert all(v == vds[0] for v in vds), "the three golden proofs must share one verifier_data"
    out = {"source": [os.path.join("near_bft_finality/proofs", d) for d in DIRS], "verifier_data": vds[0], "num_constants": 5,
           "num_sigmas": 80, "degree_bits": 12, "points": points}
    path = os.path.join(OUT, "plonky2_wrap_instance_points.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print(len(points), "points,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
