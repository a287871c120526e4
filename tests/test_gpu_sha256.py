"""`sha256_proof_u32` (near_bft_finality/src/prove_crypto/sha256.rs:62-83, tests :176-229) on the GPU: proofs of the reference's
SHA-256 circuit are accepted by the verifier restatement, carry the digest as public inputs, and can be folded by
`recursive_proof` together with their digest words as public inputs (the pattern of `prove_sub_hashes_u32`, :108-172), which
runs the in-circuit constraints of the interleave / uninterleave gates."""
import hashlib
import json

import pytest

import zklc_amd  # noqa: F401
from zklc_amd.plonky2 import HASH_GL
from zklc_amd.plonky2.recursion import RecursionProver
from zklc_amd.plonky2.sha256 import Sha256Prover
from oracle import plonky2_verifier as V, poseidon_gl as pgl

pytestmark = pytest.mark.gpu


def test_sha256_proofs_and_their_recursion(zctx):
    pgl.use_c_port()
    sp = Sha256Prover(zctx, HASH_GL)
    msgs = [bytes.fromhex("60"), bytes(range(100)), b""]
    proofs = []
    for m in msgs:
        d = hashlib.sha256(m).digest()
        (common, vd), proof = sp.sha256_proof_u32(m, d)
        V.verify(json.loads(json.dumps(proof)), vd, common)
        assert proof["public_inputs"] == [int.from_bytes(d[4 * i:4 * i + 4], "big") for i in range(8)]
        proofs.append((common, vd, proof))
    assert proofs[0][0] == proofs[2][0] and proofs[0][0] != proofs[1][0]          # one circuit per number of blocks
    with pytest.raises(AssertionError):
        sp.sha256_proof_u32(msgs[0], hashlib.sha256(b"other").digest())
    # fold two hash proofs and expose both digests (sha256.rs:131-146)
    rp = RecursionProver(zctx, HASH_GL)
    pis = proofs[0][2]["public_inputs"] + proofs[1][2]["public_inputs"]
    rc, rproof = rp.recursive_proof(proofs[0], proofs[1], pis)
    V.verify(json.loads(json.dumps(rproof)), rc.verifier_only, rc.common)
    assert rproof["public_inputs"] == pis
    bad = json.loads(json.dumps(proofs[1][2]))
    bad["public_inputs"][0] ^= 1
    with pytest.raises(AssertionError):
        rp.recursive_proof(proofs[0], (proofs[1][0], proofs[1][1], bad), pis)
    rp.close()
    sp.close()


def test_prove_header_hash_and_bp_hash(zctx):
    """header_bphash.rs:34-139 and its tests :153-221: block hash = sha256(sha256(sha256(inner_lite) || sha256(inner_rest)) ||
    prev_hash) proven as three SHA-256 proofs chained by recursion; bp_hash = sha256 of the borsh validator list (fixture C1)"""
    from conftest import load_golden, near_set_arrays
    from zklc_amd.header_bphash import BlockHashProver
    pgl.use_c_port()
    bp = BlockHashProver(zctx)
    prev_hash = hashlib.sha256(b"prev").digest()
    inner_lite = bytes((5 * i + 1) & 0xFF for i in range(208))          # INNER_LITE_BYTES (types.rs:19)
    inner_rest = bytes((3 * i + 2) & 0xFF for i in range(330))
    inner = hashlib.sha256(hashlib.sha256(inner_lite).digest() + hashlib.sha256(inner_rest).digest()).digest()
    header_hash = hashlib.sha256(inner + prev_hash).digest()
    common, vd, proof = bp.prove_header_hash(header_hash, prev_hash, inner_lite, inner_rest)
    V.verify(json.loads(json.dumps(proof)), vd, common)
    assert proof["public_inputs"] == [int.from_bytes(header_hash[4 * i:4 * i + 4], "big") for i in range(8)]
    with pytest.raises(AssertionError):
        bp.prove_header_hash(hashlib.sha256(b"not the hash").digest(), prev_hash, inner_lite, inner_rest)
    # optional extra recursion that sets caller-chosen public inputs (:97-109)
    common, vd, proof = bp.prove_header_hash(header_hash, prev_hash, inner_lite, inner_rest, public_inputs=[1, 2, 3])
    V.verify(json.loads(json.dumps(proof)), vd, common)
    assert proof["public_inputs"] == [1, 2, 3]
    _, _, validators = near_set_arrays(load_golden("ed25519_near_c1_small.json"))
    data = len(validators).to_bytes(4, "little") + b"".join(validators)
    common, vd, proof = bp.prove_bp_hash(hashlib.sha256(data).digest(), validators)
    V.verify(json.loads(json.dumps(proof)), vd, common)
    bp.close()


def test_prove_valid_keys_stakes_on_the_small_fixture(zctx):
    """keys_stakes.rs:282-343: the keys / stakes circuit proof, the SHA-256 proof of valid_keys and their recursion on fixture C1"""
    from conftest import load_golden, near_set_arrays
    from zklc_amd.keys_stakes import KeysStakesProver
    pgl.use_c_port()
    msg, approvals, validators = near_set_arrays(load_golden("ed25519_near_c1_small.json"))
    valid_keys = b"".join(bytes([pos]) + v[-48:-16] for pos, (a, v) in enumerate(zip(approvals, validators)) if len(a) == 66)
    kp = KeysStakesProver(zctx)
    common, vd, proof = kp.prove_valid_keys_stakes_in_validators_list(valid_keys, hashlib.sha256(valid_keys).digest(), validators)
    V.verify(json.loads(json.dumps(proof)), vd, common)
    assert bytes(proof["public_inputs"][:len(valid_keys)]) == valid_keys
    stake = sum(int.from_bytes(v[-16:], "little") for a, v in zip(approvals, validators) if len(a) == 66)
    assert int.from_bytes(bytes(proof["public_inputs"][len(valid_keys):]), "little") == stake
    with pytest.raises(AssertionError):
        kp.prove_valid_keys_stakes_in_validators_list(valid_keys, hashlib.sha256(b"x").digest(), validators)
    kp.close()


def test_primitive_proofs_and_the_reference_recursion_test(zctx):
    """primitives.rs tests :336-456 and recursion.rs:100-125 (`two_thirds` proof folded by `recursive_proof`)"""
    from zklc_amd.primitives import PrimitiveProver
    pgl.use_c_port()
    pp = PrimitiveProver(zctx)
    third = 0x1234567890ABCDEF
    v, v1 = 3 * third, 2 * third + 5
    tt = pp.two_thirds(v1.to_bytes(17, "little"), v.to_bytes(17, "little"))
    V.verify(json.loads(json.dumps(tt[2])), tt[1], tt[0])
    assert tt[2]["public_inputs"] == list(v1.to_bytes(17, "little"))
    with pytest.raises(AssertionError):
        pp.two_thirds((2 * third - 5).to_bytes(17, "little"), v.to_bytes(17, "little"))
    ch = pp.prove_consecutive_heights((105971807).to_bytes(8, "little"), (105971806).to_bytes(8, "little"))
    V.verify(json.loads(json.dumps(ch[2])), ch[1], ch[0])
    eq = pp.prove_eq_array(bytes(range(32)), bytes(range(32)))
    V.verify(json.loads(json.dumps(eq[2])), eq[1], eq[0])
    rp = RecursionProver(zctx, HASH_GL)
    rc, rproof = rp.recursive_proof(tt)
    V.verify(json.loads(json.dumps(rproof)), rc.verifier_only, rc.common)
    bad = json.loads(json.dumps(tt[2]))
    bad["public_inputs"][-1] = 10000          # recursion.rs:152
    with pytest.raises(AssertionError):
        rp.recursive_proof((tt[0], tt[1], bad))
    rp.close()
    pp.close()
