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
