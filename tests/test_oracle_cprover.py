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


@pytest.mark.parametrize("shape", ["recursion", "ed25519"])
def test_c_prover_every_gate_type_bytes_equal_the_python_restatement(shape):
    """the synthetic circuits of the GPU parity tests (every gate type of the two reference circuit shapes, 19 in total, random
    satisfying witnesses, 2^6 rows): the C prover's bytes are the Python restatement's"""
    from zklc_amd.plonky2 import synthetic as SY, gates as G, standard_recursion_config, wide_ecc_config
    pgl.use_c_port()
    if shape == "recursion":
        cfg = standard_recursion_config()
        mix = SY.recursion_shape_mix(cfg) + [(G.ExponentiationGate(20), 3)]
    else:
        cfg = wide_ecc_config()
        mix = SY.ed25519_shape_mix(cfg)
    data, wires, pis = SY.synthetic_circuit(6, cfg, mix, num_public_inputs=11, seed=3)
    common = data.common_data()
    oproof, ovd = OP.prove(common, data.constants, data.sigmas, wires, pis, V.HasherGL)
    got, _, vd = cport.plonky2_prove(data, wires, pis, nthreads=2, verifier_data=True)
    assert got == S.proof_to_bytes(json.loads(json.dumps(oproof)), common, S.HASH_GL)
    assert vd == json.loads(json.dumps(ovd))


def test_c_prover_refuses_unknown_gate_codes(inner):
    data, wires, pis, common, proof, vd = inner

    class Fake:
        pass
    fake = Fake()
    fake.__dict__.update(data.__dict__)
    g0 = Fake()
    g0.code, g0.params = 99, [0, 0, 0, 0]
    fake.gates = [g0] + list(data.gates[1:])
    with pytest.raises(ValueError):
        cport.plonky2_prove(fake, wires, pis)
