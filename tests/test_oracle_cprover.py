"""The oracle's complete C + OpenMP prover (oracle/c/plonky2_prover_oracle.c: the CPU baseline of bench.py) against the Python
restatement (oracle/plonky2_prover.py) and the verifier restatement (oracle/plonky2_verifier.py, pinned by the reference's golden
proofs): identical proof bytes on a small circuit, independent of the thread count; a complete proof of a recursion circuit
(the in-circuit verifier of two inner proofs: 13 gate types, 2^13 rows x 135 wires) accepted by the verifier; circuits outside
the recursion gate set refused; an unsatisfied witness reported."""
import json

import numpy as np
import pytest

from oracle import cport, plonky2_prover as OP, plonky2_verifier as V, poseidon_gl as pgl
from zklc_amd.plonky2 import recursion as R, serialization as S
from test_recursion import _inner_circuit


@pytest.fixture(scope="module")
def inner():
    pgl.use_c_port()
    data, wires, pis = _inner_circuit(64)
    common = data.common_data()
    proof, vd = OP.prove(common, data.constants, data.sigmas, wires, pis, V.HasherGL)
    return data, wires, pis, common, json.loads(json.dumps(proof)), json.loads(json.dumps(vd))


def test_c_prover_bytes_equal_the_python_restatement(inner):
    data, wires, pis, common, proof, vd = inner
    want = S.proof_to_bytes(proof, common, S.HASH_GL)
    got1, secs, vd_c = cport.plonky2_prove(data, wires, pis, nthreads=1, verifier_data=True)
    got4, _ = cport.plonky2_prove(data, wires, pis, nthreads=4)
    assert got1 == want and got4 == want
    assert vd_c == vd
    assert set(secs) == set(cport.PLONKY2_PROVE_STAGES) and secs["proof"] > 0
    # a witness that does not satisfy the copy constraints: Z does not close
    bad = np.array(wires, dtype=np.uint64).copy()
    bad[0, 3] ^= np.uint64(1)
    with pytest.raises(AssertionError):
        cport.plonky2_prove(data, bad, pis)


def test_c_prover_on_a_recursion_circuit_is_accepted_by_the_verifier(inner):
    data, wires, pis, common, proof, vd = inner
    rdata, targets = R.recursive_circuit([common, common], 0)
    rc = R.RecursiveCircuit(rdata, targets, None, [common, common], 0)
    raw = S.proof_to_bytes(proof, common, S.HASH_GL)
    rc.compile(R.recursive_witness(targets, [(proof, vd), (proof, vd)], []), [raw, raw])
    w, p = rdata.generate_witness_native(None, input_values=rc.input_vector([(vd, raw), (vd, raw)], [])[None, :])
    got, secs, rvd = cport.plonky2_prove(rdata, w[0], [int(x) for x in p[0]], verifier_data=True)
    rcommon = rdata.common_data()
    assert len(got) == S.proof_size(rcommon, S.HASH_GL)
    V.verify(json.loads(json.dumps(S.proof_from_bytes(got, rcommon, S.HASH_GL))), rvd, rcommon)
    print("complete C proof of a 2^%d x 135 recursion circuit: %s" % (rdata.degree_bits, {k: round(v, 3) for k, v in secs.items()}))
    # a tampered proof must not pass (the verifier, not the prover, is what this checks)
    t = bytearray(got)
    t[40] ^= 1
    with pytest.raises(AssertionError):
        V.verify(json.loads(json.dumps(S.proof_from_bytes(bytes(t), rcommon, S.HASH_GL))), rvd, rcommon)


def test_c_prover_refuses_gates_outside_the_recursion_set():
    from zklc_amd.plonky2 import CircuitBuilder, wide_ecc_config
    from zklc_amd.plonky2 import ed25519_circuit as E
    b = CircuitBuilder(wide_ecc_config())
    g = E.Gadgets(b)
    x, y = g.virtual_biguint(2), g.virtual_biguint(2)
    g.add_biguint(x, y) if hasattr(g, "add_biguint") else g.add_nonnative(g.virtual_biguint(8), g.virtual_biguint(8))
    data = b.build()
    assert any(gt.code > 13 for gt in data.gates)
    wires = np.zeros((data.config["num_wires"], data.n), dtype=np.uint64)
    with pytest.raises(ValueError):
        cport.plonky2_prove(data, wires, [0] * data.num_public_inputs)
