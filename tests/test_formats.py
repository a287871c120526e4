"""the formats either side of the proving path (zklc_amd/formats.py), against the files the reference holds."""
import json

import pytest

from conftest import load_golden
from zklc_amd import formats as F
from zklc_amd.plonky2 import serialization as S, HASH_GL, HASH_BN128


def test_verifier_data_bin_is_rewritten_byte_for_byte():
    """near_bft_finality/proofs/*/verifier_data.bin (VerifierCircuitData::to_bytes with the default gate serializer, written by
    bin/prove_block.rs:320-458) from the verifier_data.json / common_data.json next to it: all three golden files"""
    g = load_golden("plonky2_verifier_data_bins.json")
    assert len(g["cases"]) == 3
    for c in g["cases"]:
        want = bytes.fromhex(c["verifier_data_bin"])
        got = F.verifier_data_to_bytes(c["verifier_only"], c["common_data"], HASH_BN128)
        assert got == want, c["name"]
        vo = F.verifier_only_to_bytes(c["verifier_only"], HASH_BN128)
        assert want.startswith(vo) and len(vo) == 8 + 32 * 17
        assert F.verifier_only_from_bytes(vo, HASH_BN128) == json.loads(json.dumps(c["verifier_only"]))
    with pytest.raises(ValueError):
        F.verifier_only_from_bytes(vo[:-1], HASH_BN128)
    bad = dict(g["cases"][0]["common_data"], gates=["U32AddManyGate { num_addends: 3, num_ops: 5 }"])
    with pytest.raises(ValueError, match="DefaultGateSerializer"):
        F.common_data_to_bytes(bad)


def test_nats_task_messages_round_trip():
    """InputTask / OutputTask (near_bft_finality/src/types.rs:172-192) as prove_approvals_with_client exchanges them
    (signatures.rs:188-230): serde_json of Vec<u8> fields = arrays of numbers"""
    j = load_golden("ed25519_near_c1_small.json")
    msg = bytes.fromhex(j["msg"])
    e = j["entries"][1]
    sig, pk = bytes.fromhex(e["approval"])[2:], bytes.fromhex(e["validator_tail"])[1:33]
    raw = F.input_task_to_json(msg, sig, pk, 1)
    d = json.loads(raw)
    assert list(d) == ["message", "approval", "validator", "signature_index"]           # serde field order
    assert d["message"] == list(msg) and len(d["approval"]) == 64 and len(d["validator"]) == 32 and d["signature_index"] == 1
    assert F.input_task_from_json(raw) == {"message": msg, "approval": sig, "validator": pk, "signature_index": 1}
    with pytest.raises(ValueError):
        F.input_task_to_json(msg, sig[:-1], pk, 0)
    g = load_golden("plonky2_near_random_CGZP.json")
    proof_bin = open(__import__("os").path.join(__import__("conftest").GOLDEN, "plonky2_near_random_CGZP_proof.bin"), "rb").read()
    out = F.output_task_to_json(proof_bin, g["verifier_data"], 7, HASH_BN128)
    back = F.output_task_from_json(out, HASH_BN128)
    assert back["proof"] == proof_bin and back["signature_index"] == 7
    assert back["verifier_only"] == json.loads(json.dumps(g["verifier_data"]))
    # the proof bytes inside are what ProofWithPublicInputs::from_bytes takes (signatures.rs:225-228)
    pj = S.proof_from_bytes(back["proof"], g["common_data"], HASH_BN128)
    assert pj["public_inputs"] == g["proof"]["public_inputs"]


def test_gnark_json_schema_round_trips():
    """the JSON files the Go wrap reads (gnark-plonky2-verifier/types/deserialize.go, variables/deserialize.go; testdata
    test_circuit/*.json): JSON -> ProofWithPublicInputs bytes -> JSON is the identity on every field the fixture keeps"""
    g = load_golden("plonky2_gnark_test_circuit.json")
    common = g["common_data"]
    pj = g["proof"]
    full_rounds = common["fri_params"]["config"]["num_query_rounds"]
    kept = len(pj["proof"]["opening_proof"]["query_round_proofs"])
    # the fixture keeps only the first query rounds: serialise what is there by repeating the last kept round
    padded = json.loads(json.dumps(pj))
    qr = padded["proof"]["opening_proof"]["query_round_proofs"]
    padded["proof"]["opening_proof"]["query_round_proofs"] = qr + [qr[-1]] * (full_rounds - kept)
    raw = S.proof_to_bytes(padded, common, HASH_BN128)
    assert len(raw) == S.proof_size(common, HASH_BN128)
    back = S.proof_from_bytes(raw, common, HASH_BN128)
    assert json.loads(json.dumps(back)) == padded


def test_groth16_encodings_against_the_reference_proof():
    """contracts/hardhat/test/proof_with_witness.json: raw 256-byte form (gnark WriteRawTo, web-api.go:90-98), the JSON the web API
    answers with, and Verifier.sol's compressProof / decompression (:201-365,427-449): the compressed proof decompresses to the
    same eight words, the points are on the curve (the square roots exist), tampering is rejected"""
    k = load_golden("groth16_kat.json")
    proof = [int(x) for x in k["proof"]]
    raw = F.proof_to_raw_bytes(proof)
    assert len(raw) == 256 and F.proof_from_raw_bytes(raw) == proof
    assert F.proof_to_web_api_json(proof, k["inputs"]) == {"inputs": k["inputs"], "proof": k["proof"]}
    c = F.compress_proof(proof)
    assert len(c) == 4 and all(0 <= x < 1 << 256 for x in c)
    assert c[0] >> 1 == proof[0] and c[3] >> 1 == proof[6] and c[2] >> 2 == proof[3] and c[1] == proof[2]
    assert F.decompress_proof(c) == proof
    # the sign bit selects the other root: flipping it yields the negated point, still a valid encoding
    neg = F.decompress_proof([c[0] ^ 1, c[1], c[2], c[3]])
    assert neg[0] == proof[0] and neg[1] == F.P_BN254 - proof[1]
    with pytest.raises(F.ProofInvalid):
        F.compress_proof([proof[0], proof[1] ^ 1] + proof[2:])                 # A is not on the curve
    with pytest.raises(F.ProofInvalid):
        F.compress_proof(proof[:2] + [proof[2] ^ 1] + proof[3:])               # B is not on the curve
    with pytest.raises(F.ProofInvalid):
        F.compress_g1(F.P_BN254, 1)
    assert F.compress_g1(0, 0) == 0 and F.decompress_g1(0) == (0, 0)
    assert F.compress_g2(0, 0, 0, 0) == (0, 0) and F.decompress_g2(0, 0) == (0, 0, 0, 0)
    # the verification key of Verifier.sol holds G2 points too (-beta, -gamma, -delta): they compress and come back
    vk = {n: int(v) for n, v in k["vk"].items()}
    for name in ("BETA_NEG", "GAMMA_NEG", "DELTA_NEG"):
        pt = (vk[name + "_X_0"], vk[name + "_X_1"], vk[name + "_Y_0"], vk[name + "_Y_1"])
        assert F.decompress_g2(*F.compress_g2(*pt)) == pt
    for name in ("ALPHA", "CONSTANT", "PUB_0", "PUB_3"):
        pt = (vk[name + "_X"], vk[name + "_Y"])
        assert F.decompress_g1(F.compress_g1(*pt)) == pt
