"""GPU plonky2 prover (zklc_plonky2_prove) against the oracle: the proof bytes must equal the pure-Python prover
restatement's proof bit for bit, and the verifier restatement (pinned by the reference's golden proofs) must accept it."""
import json

import numpy as np
import pytest

import zklc_amd
from zklc_amd.plonky2 import CircuitBuilder, serialization as S, HASH_GL, HASH_BN128
from oracle import plonky2_prover as OP, plonky2_verifier as V

pytestmark = pytest.mark.gpu
HASHERS = {HASH_GL: V.HasherGL, HASH_BN128: V.HasherBN128}


def small_circuit(n_extra_rows=0):
    b = CircuitBuilder()
    x = b.add_virtual_public_input()
    y = b.add_virtual_target()
    z = b.mul(x, y)
    w = b.add(z, b.constant(5))
    b.split_le(x, 10)
    lo, hi = b.mul_add_u32(x, y, b.constant(77))
    s = b.sub(w, z)
    b.connect(s, b.constant(5))
    b.register_public_input(lo)
    acc = z
    for i in range(n_extra_rows * 20):
        acc = b.mul_add(acc, y, x)
    data = b.build()
    wires, pis = data.generate_witness({x: 1000, y: 4000000000})
    return data, wires, pis


@pytest.mark.parametrize("hasher", [HASH_GL, HASH_BN128])
@pytest.mark.parametrize("extra", [0, 40])
def test_proof_matches_oracle_bit_for_bit(zctx, hasher, extra):
    data, wires, pis = small_circuit(extra)
    common = data.common_data()
    H = HASHERS[hasher]
    prover = data.prover(zctx, hasher)
    proof_bytes = prover.prove_bytes(wires, pis)
    assert len(proof_bytes) == S.proof_size(common, hasher)
    proof = S.proof_from_bytes(proof_bytes, common, hasher)
    vd = prover.verifier_data()
    trace = {}
    oproof, ovd = OP.prove(common, data.constants, data.sigmas, wires, pis, H, trace=trace)
    assert vd == ovd, "constants_sigmas cap / circuit digest"
    ch = prover.last_challenges()
    nch = 2
    assert ch[:nch] == trace["betas"] and ch[nch:2 * nch] == trace["gammas"], "betas/gammas (wires commitment)"
    assert ch[2 * nch:3 * nch] == trace["alphas"], "alphas (Z / partial products commitment)"
    assert tuple(ch[3 * nch:3 * nch + 2]) == trace["zeta"], "zeta (quotient commitment)"
    assert tuple(ch[3 * nch + 2:3 * nch + 4]) == trace["fri_alpha"], "fri alpha (openings)"
    assert proof_bytes == S.proof_to_bytes(oproof, common, hasher), "proof bytes"
    V.verify(json.loads(json.dumps(proof)), vd, common)
    # proving twice gives the same bytes (no hidden state)
    assert prover.prove_bytes(wires, pis) == proof_bytes


def test_bad_witness_is_rejected(zctx):
    data, wires, pis = small_circuit()
    prover = data.prover(zctx, HASH_GL)
    bad = wires.copy()
    bad[0, 1] ^= 1    # break a copy-constrained cell
    with pytest.raises(zklc_amd.ZklcError):
        prover.prove_bytes(bad, pis)
