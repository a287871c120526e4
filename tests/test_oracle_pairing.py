"""The pairing oracle is pinned by the reference's Groth16 known-answer proof (Verifier.sol verifying key +
proof_with_witness.json) and rejects the reference's tampered vectors (verify.ts); bilinearity on top."""
import pytest

from conftest import load_golden
from oracle import bn254 as B
from oracle import bn254_pairing as PR


def kat_vk(j):
    v = {k: int(x) for k, x in j["vk"].items()}
    g2 = lambda n: ((v[n + "_NEG_X_0"], v[n + "_NEG_X_1"]), (v[n + "_NEG_Y_0"], v[n + "_NEG_Y_1"]))
    return {"alpha": (v["ALPHA_X"], v["ALPHA_Y"]), "beta_neg": g2("BETA"), "gamma_neg": g2("GAMMA"), "delta_neg": g2("DELTA"),
            "ic": [(v["CONSTANT_X"], v["CONSTANT_Y"])] + [(v["PUB_%d_X" % i], v["PUB_%d_Y" % i]) for i in range(4)]}


def test_verifying_key_points_are_on_their_curves():
    vk = kat_vk(load_golden("groth16_kat.json"))
    assert B.is_on_curve(vk["alpha"]) and all(B.is_on_curve(p) for p in vk["ic"])
    for k in ("beta_neg", "gamma_neg", "delta_neg"):
        assert B.g2_is_on_curve(vk[k]) and B.g2_mul(B.R, vk[k]) is None


def test_reference_groth16_proof_verifies_and_tampered_vectors_do_not():
    j = load_golden("groth16_kat.json")
    vk = kat_vk(j)
    proof, inputs = [int(x) for x in j["proof"]], [int(x) for x in j["inputs"]]
    assert PR.groth16_verify(vk, proof, inputs)
    assert not PR.groth16_verify(vk, proof, [int(x) for x in j["incorrect_inputs"]])
    assert not PR.groth16_verify(vk, [int(x) for x in j["incorrect_proof"]], inputs)
    bad = list(inputs)
    bad[2] += 1
    assert not PR.groth16_verify(vk, proof, bad)


def test_bilinearity_and_non_degeneracy():
    e = PR.pairing(B.G1, B.G2)
    assert e != PR.F12_ONE and PR.f12_pow(e, B.R) == PR.F12_ONE
    a, b = 0x1234567, 0x89abcdef
    assert PR.pairing(B.mul(a, B.G1), B.g2_mul(b, B.G2)) == PR.f12_pow(e, a * b % B.R)
    assert PR.pairing_check([(B.mul(a, B.G1), B.G2), (B.neg(B.G1), B.g2_mul(a, B.G2))])
    assert PR.pairing_check([(None, B.G2)]) and not PR.pairing_check([(B.G1, B.G2)])
