"""`recursive_proof` on the GPU (near_bft_finality/src/prove_crypto/recursion.rs:16-97 and its tests :100-158): the outer
proofs of the in-circuit verifier are produced by the HIP prover and accepted by the verifier restatement (oracle, pinned by
the reference's golden proofs)."""
import json

import pytest

import zklc_amd  # noqa: F401
from zklc_amd.plonky2 import HASH_BN128, HASH_GL
from zklc_amd.plonky2.recursion import RecursionProver
from oracle import plonky2_verifier as V, poseidon_gl as pgl
from test_recursion import _inner_circuit

pytestmark = pytest.mark.gpu


def test_recursive_proof_valid_fold_wrap_and_invalid(zctx):
    pgl.use_c_port()
    data, wires, pis = _inner_circuit(64)
    common = data.common_data()
    inner = data.prover(zctx, HASH_GL)
    proof = inner.prove(wires, pis)
    vd = inner.verifier_data()
    V.verify(json.loads(json.dumps(proof)), vd, common)
    rp = RecursionProver(zctx, HASH_GL)
    # recursion.rs:120-125: one inner proof, no public inputs
    rc1, p1 = rp.recursive_proof((common, vd, proof))
    V.verify(json.loads(json.dumps(p1)), rc1.verifier_only, rc1.common)
    assert p1["public_inputs"] == []
    # the fold step of signatures.rs:97-105: a recursion proof and a fresh inner proof, then public inputs (:131-139)
    rc2, p2 = rp.recursive_proof((rc1.common, rc1.verifier_only, p1), (common, vd, proof))
    V.verify(json.loads(json.dumps(p2)), rc2.verifier_only, rc2.common)
    rc3, p3 = rp.recursive_proof((rc2.common, rc2.verifier_only, p2), None, [5, 6, 7])
    V.verify(json.loads(json.dumps(p3)), rc3.verifier_only, rc3.common)
    assert p3["public_inputs"] == [5, 6, 7]
    # the circuit of a shape is built once and reused
    rc1b, p1b = rp.recursive_proof((common, vd, proof))
    assert rc1b is rc1 and p1b == p1
    # Poseidon-BN128 outer configuration (the wrap)
    wrap = RecursionProver(zctx, HASH_BN128)
    rcw, pw = wrap.recursive_proof((rc3.common, rc3.verifier_only, p3))
    V.verify(json.loads(json.dumps(pw)), rcw.verifier_only, rcw.common)
    # recursion.rs:127-158: a modified public input of the inner proof -> no witness
    bad = json.loads(json.dumps(proof))
    bad["public_inputs"][-1] = 10000
    with pytest.raises(AssertionError):
        rp.recursive_proof((common, vd, bad))
    # a recursion proof does not verify under the verifier data of another circuit
    with pytest.raises(AssertionError):
        rp.recursive_proof((rc1.common, rc2.verifier_only, p1))
    # the opt-in device interpreter for the recursion circuit's witness: the same proof bytes from the matrix in HBM
    rpd = RecursionProver(zctx, HASH_GL, device_witness=True)
    rc2d, p2d = rpd.recursive_proof((rc1.common, rc1.verifier_only, p1), (common, vd, proof), raw=True)
    _, p2h = rp.recursive_proof((rc1.common, rc1.verifier_only, p1), (common, vd, proof), raw=True)
    assert bytes(p2d) == bytes(p2h)
    with pytest.raises(AssertionError):
        rpd.recursive_proof((common, vd, bad))
    rpd.close()
    inner.close()
    rp.close()
    wrap.close()
