"""The reference's golden plonky2 proofs verify through this repo's oracle stack (transcript, PoW, Poseidon-BN254
Merkle paths to the caps, evaluation-domain order, FRI folding): pins oracle/poseidon_gl.py,
oracle/poseidon_bn254.py and the Merkle conventions the GPU kernels are tested against."""
import copy
import glob
import json
import os

import pytest

from conftest import load_golden
from oracle import plonky2_verifier as V

FIX = ["plonky2_near_random_CGZP.json", "plonky2_gnark_test_circuit.json"]


@pytest.mark.parametrize("name", FIX)
def test_golden_proof_verifies(name):
    j = load_golden(name)
    ch = V.verify(j["proof"], j["verifier_data"], j["common_data"])
    assert len(ch["query_indices"]) == 28 and len(ch["fri_betas"]) == 2
    assert ch["pow_response"] < 1 << 48


@pytest.mark.parametrize("what", ["pow", "sibling", "leaf", "step_eval", "final_poly", "cap", "opening"])
def test_tampering_is_detected(what):
    j = copy.deepcopy(load_golden(FIX[0]))
    op = j["proof"]["proof"]["opening_proof"]
    q0 = op["query_round_proofs"][0]
    if what == "pow":
        op["pow_witness"] += 1
    elif what == "sibling":
        s = q0["initial_trees_proof"]["evals_proofs"][1][1]["siblings"]
        s[3] = str(int(s[3]) ^ 1)
    elif what == "leaf":
        q0["initial_trees_proof"]["evals_proofs"][1][0][7] ^= 1
    elif what == "step_eval":
        q0["steps"][1]["evals"][5][0] ^= 1
    elif what == "final_poly":
        op["final_poly"]["coeffs"][2][0] ^= 1
    elif what == "cap":
        c = j["proof"]["proof"]["wires_cap"]
        c[0] = str(int(c[0]) ^ 1)
    elif what == "opening":
        j["proof"]["proof"]["openings"]["wires"][3][0] ^= 1
    with pytest.raises(AssertionError):
        V.verify(j["proof"], j["verifier_data"], j["common_data"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_all_reference_proofs_full_28_rounds():
    R = "/root/reference/"
    cases = []
    for d in sorted(glob.glob(R + "near_bft_finality/proofs/*/*/")):
        if os.path.exists(d + "proof.json"):
            cases.append((d + "proof.json", d + "verifier_data.json", d + "common_data.json"))
    t = R + "gnark-plonky2-verifier/testdata/test_circuit/"
    cases.append((t + "proof_with_public_inputs.json", t + "verifier_only_circuit_data.json", t + "common_circuit_data.json"))
    assert len(cases) == 4
    for pj, vj, cj in cases:
        V.verify(json.load(open(pj)), json.load(open(vj)), json.load(open(cj)))
