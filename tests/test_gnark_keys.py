"""gnark's pk.bin / vk.bin (zklc_amd/gnark_keys.py; gnark-plonky2-verifier/verifier/util.go:172-216 writes them, :337-389 reads them).
The reference holds no key file, so these are structure and round-trip tests (the module header says PARITY UNPINNED): a key made
by the oracle's setup survives write -> read in the raw and the compressed encodings, with points at infinity, an opaque trailer
and every malformed input class refused."""
import numpy as np
import pytest

from oracle import groth16 as G, bn254 as B
from zklc_amd import gnark_keys as K
from zklc_amd.formats import ProofInvalid, P_BN254


def _keys(n_con=12, n_pub=2):
    r1cs, wit = G.square_chain_r1cs(n_con, n_public=n_pub)
    pk, vk = G.setup(r1cs, n_pub, (0x1234567891, 0xabcdef12345, 0x777766665555, 0x3133731337, 0x42424242))
    vk = dict(vk)
    vk.setdefault("beta1", pk["beta1"])
    vk.setdefault("delta1", pk["delta1"])
    return pk, vk, n_pub


@pytest.mark.parametrize("raw", [True, False])
def test_verifying_key_round_trip(raw):
    pk, vk, _ = _keys()
    data = K.vk_to_gnark_bytes(vk, raw=raw, trailer=b"\x00\x00\x00\x00tail")
    n_k = len(vk["K"])
    assert len(data) == (3 * 64 + 3 * 128 + 4 + n_k * 64 if raw else 3 * 32 + 3 * 64 + 4 + n_k * 32) + 8
    got = K.vk_from_gnark_bytes(data, raw=raw)
    assert got["trailer"] == b"\x00\x00\x00\x00tail"
    for name in ("alpha1", "beta1", "beta2", "gamma2", "delta1", "delta2", "K"):
        assert got[name] == vk[name], name


@pytest.mark.parametrize("raw", [True, False])
def test_proving_key_round_trip_with_points_at_infinity(raw):
    pk, _, n_pub = _keys(n_con=20)
    assert any(p is None for p in pk["A"]) or any(p is None for p in pk["B1"])
    data = K.pk_to_gnark_bytes(pk, raw=raw)
    got = K.pk_from_gnark_bytes(data, n_pub, raw=raw)
    assert got["n"] == pk["n"] and got["trailer"] == b""
    assert list(got["infinity_a"]) == [p is None for p in pk["A"]]
    assert list(got["infinity_b"]) == [p is None for p in pk["B1"]]
    assert got["A"] == [p for p in pk["A"] if p is not None]
    assert got["B1"] == [p for p in pk["B1"] if p is not None]
    assert got["B2"] == [p for p in pk["B2"] if p is not None]
    for name in ("alpha1", "beta1", "delta1", "beta2", "delta2", "K"):
        assert got[name] == pk[name], name
    assert got["Z"] == pk["Z"][:pk["n"] - 1]
    # the domain header: cardinality and a primitive root of that order
    assert int.from_bytes(data[:8], "big") == pk["n"]
    gen = int.from_bytes(data[40:72], "big")
    assert pow(gen, pk["n"], G.R) == 1 and pow(gen, pk["n"] // 2, G.R) != 1


def test_point_encodings():
    g1, g2 = B.mul(5, B.G1), B.g2_mul(7, B.G2)
    for pt in (g1, B.neg(g1)):
        for raw in (True, False):
            enc = K.write_g1(pt, raw)
            assert len(enc) == (64 if raw else 32) and K.read_g1(K._Reader(enc)) == pt
    for pt in (g2, B.g2_neg(g2)):
        for raw in (True, False):
            enc = K.write_g2(pt, raw)
            assert len(enc) == (128 if raw else 64) and K.read_g2(K._Reader(enc)) == pt
    # the raw G2 layout is the one the reference's web-api slices into Verifier.sol's words: X.A1, X.A0, Y.A1, Y.A0
    enc = K.write_g2(g2, True)
    assert int.from_bytes(enc[:32], "big") == g2[0][1] and int.from_bytes(enc[96:], "big") == g2[1][0]
    # compressed flags: 0b10 = the smaller y, 0b11 = the larger one
    a, b = K.write_g1(g1, False)[0] >> 6, K.write_g1(B.neg(g1), False)[0] >> 6
    assert {a, b} == {2, 3} and (a == 3) == (g1[1] > (P_BN254 - 1) // 2)
    assert K.read_g1(K._Reader(K.write_g1(None, False))) is None
    assert K.points_to_words([g1, None]).tolist()[1] == [0] * 8
    assert [int(v) for v in K.points_to_words([g2], g2=True)[0]] == B.g2_to_words(g2)


def test_malformed_keys_are_refused():
    pk, vk, n_pub = _keys()
    good = K.vk_to_gnark_bytes(vk)
    with pytest.raises(ProofInvalid):
        K.vk_from_gnark_bytes(good[:100])
    bad = bytearray(good)
    bad[63] ^= 1                                      # alpha1.y: off the curve
    with pytest.raises(ProofInvalid):
        K.vk_from_gnark_bytes(bytes(bad))
    bad = bytearray(good)
    bad[0:32] = P_BN254.to_bytes(32, "big")           # x = p: not reduced (and the flag bits read as "uncompressed")
    with pytest.raises(ProofInvalid):
        K.vk_from_gnark_bytes(bytes(bad))
    data = bytearray(K.pk_to_gnark_bytes(pk))
    data[7] ^= 1                                      # cardinality no power of two
    with pytest.raises(ProofInvalid):
        K.pk_from_gnark_bytes(bytes(data), n_pub)
    with pytest.raises(ProofInvalid):
        K.pk_from_gnark_bytes(K.pk_to_gnark_bytes(pk), n_pub + 1)     # K does not match nbWires - nbPublic
    with pytest.raises(ProofInvalid):
        K.pk_from_gnark_bytes(K.pk_to_gnark_bytes(pk)[:-3], n_pub)
