"""oracle/groth16.py (gnark's Groth16 over BN254 restated: setup, computeH, the five multi-exponentiations, verification) on a
small satisfiable system: proofs verify through the KAT-pinned verification equation (oracle.bn254_pairing.groth16_verify),
wrong public inputs / tampered proofs / a wrong witness do not; the quotient really divides (h has degree <= n - 2)."""
import pytest

from oracle import groth16 as G, bn254 as B, bn254_fr as FR


@pytest.fixture(scope="module")
def system():
    r1cs, wit = G.square_chain_r1cs(6, n_public=2)
    pk, vk = G.setup(r1cs, 2, (0x1234567, 0xabcdef1, 0x7777, 0x31337, 0x424242))
    return r1cs, wit, pk, vk


def test_quotient_is_a_polynomial_only_for_a_satisfying_witness(system):
    r1cs, wit, pk, vk = system
    w = wit([5, 9], 3)
    a, b, c = G.abc_evaluations(r1cs, w, pk["n"])
    h = G.compute_h(a, b, c)
    assert h[pk["n"] - 1] == 0                      # deg h <= n - 2
    # A B - C = h Z at a point outside the domain
    x = 0x1d3f
    ca, cb, cc = (FR.ntt(v, inverse=True) for v in (a, b, c))
    ev = lambda co: sum(k * pow(x, i, G.R) for i, k in enumerate(co)) % G.R
    assert (ev(ca) * ev(cb) - ev(cc)) % G.R == ev(h) * (pow(x, pk["n"], G.R) - 1) % G.R
    bad = list(w)
    bad[-1] = (bad[-1] + 1) % G.R
    a, b, c = G.abc_evaluations(r1cs, bad, pk["n"])
    h = G.compute_h(a, b, c)
    ca, cb, cc = (FR.ntt(v, inverse=True) for v in (a, b, c))
    assert (ev(ca) * ev(cb) - ev(cc)) % G.R != ev(h) * (pow(x, pk["n"], G.R) - 1) % G.R


def test_proofs_verify_through_the_pinned_equation(system):
    r1cs, wit, pk, vk = system
    w = wit([5, 9], 3)
    proof = G.prove(pk, r1cs, w, 0x1111, 0x2222)
    assert B.is_on_curve(proof[0]) and B.g2_is_on_curve(proof[1]) and B.is_on_curve(proof[2])
    assert G.verify(vk, proof, [5, 9])
    assert not G.verify(vk, proof, [5, 10])
    assert not G.verify(vk, (proof[0], proof[1], B.add(proof[2], B.G1)), [5, 9])
    # another randomisation of the same statement is a different, valid proof (the reason proof bytes cannot be golden)
    p2 = G.prove(pk, r1cs, w, 0x3333, 0x4444)
    assert p2 != proof and G.verify(vk, p2, [5, 9])
    w8 = G.proof_to_uint256x8(proof)
    assert len(w8) == 8 and w8[0] == proof[0][0] and w8[2] == proof[1][0][1]
