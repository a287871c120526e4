"""`groth16.Prove` assembled on the GPU (zklc_amd/groth16.py: computeH as seven NTTs + one pointwise kernel in HBM, Ar / Bs / Bs1 /
Krs as four MSMs) against oracle/groth16.py (gnark's algorithm restated in Python) on the same key, witness and blinding scalars:
the eight proof words must be IDENTICAL, the proof must pass the GPU pairing check under the verifying key and the KAT-pinned
verifier restatement, and survive the 256-byte / compressed encodings of the contracts (gnark-plonky2-verifier/cmd/web-api.go:77-98,
contracts/hardhat/contracts/Verifier.sol:427-449)."""
import numpy as np
import pytest

from oracle import groth16 as G, bn254 as B
from zklc_amd import formats as F
from zklc_amd.groth16 import Groth16Prover, g1_words, g2_words

pytestmark = pytest.mark.gpu


def test_groth16_prove_on_the_gpu_equals_the_oracle_and_verifies(zctx):
    n_con, n_pub = 100, 3
    r1cs, wit = G.square_chain_r1cs(n_con, n_public=n_pub)
    pk, vk = G.setup(r1cs, n_pub, (0x1234567891, 0xabcdef12345, 0x777766665555, 0x3133731337, 0x42424242))
    assert pk["n"] == 128
    prover = Groth16Prover(zctx, pk)
    for seed, (r, s) in enumerate([(0x1111222233334444, 0x5555666677778888), (G.R - 2, 3)]):
        pubs = [11 + seed, 22, 33]
        w = wit(pubs, 7 + seed)
        abc = G.abc_evaluations(r1cs, w, pk["n"])
        got = prover.prove(w, abc, r, s)
        want = G.proof_to_uint256x8(G.prove(pk, r1cs, w, r, s))
        assert got == want, "GPU Groth16 proof differs from the oracle prover"
        print("groth16 prove (n = 128): ms", prover.last_ms)
        assert G.verify(vk, ((got[0], got[1]), ((got[3], got[2]), (got[5], got[4])), (got[6], got[7])), pubs)
        # the verification equation on the GPU pairing kernel: e(A, B) e(C, -delta) e(alpha, -beta) e(L, -gamma) = 1
        l = vk["K"][0]
        for x, pt in zip(pubs, vk["K"][1:]):
            l = B.add(l, B.mul(x, pt))
        a, c = (got[0], got[1]), (got[6], got[7])
        b = ((got[3], got[2]), (got[5], got[4]))
        g1 = np.array([[g1_words(a), g1_words(c), g1_words(vk["alpha1"]), g1_words(l)]], dtype=np.uint64)
        g2 = np.array([[g2_words(b), g2_words(B.g2_neg(vk["delta2"])), g2_words(B.g2_neg(vk["beta2"])), g2_words(B.g2_neg(vk["gamma2"]))]],
                      dtype=np.uint64)
        ok, _ = zctx.bn254_pairing_check(g1, g2, 4)
        assert int(ok[0]) == 1
        bad = g1.copy()
        bad[0, 3] = np.array(g1_words(B.add(l, B.G1)), dtype=np.uint64)
        ok, _ = zctx.bn254_pairing_check(bad, g2, 4)
        assert int(ok[0]) == 0
        raw = F.proof_to_raw_bytes(got)
        assert len(raw) == 256 and F.proof_from_raw_bytes(raw) == got
        assert F.decompress_proof(F.compress_proof(got)) == got
    # a witness that does not satisfy the system yields a proof that does not verify (gnark's prover does not check either)
    w = wit([1, 2, 3], 5)
    w[-1] = (w[-1] + 1) % G.R
    got = prover.prove(w, G.abc_evaluations(r1cs, w, pk["n"]), 5, 6)
    assert not G.verify(vk, ((got[0], got[1]), ((got[3], got[2]), (got[5], got[4])), (got[6], got[7])), [1, 2, 3])


def test_groth16_prove_over_fixed_base_tables_equals_the_oracle(zctx, monkeypatch):
    """ZKLC_GROTH16_FIXED=1: the four big sums over fixed-base tables (built once per key); the proof words are the oracle's and the
    plain form's -- the affine result of a multi-exponentiation is canonical"""
    n_con, n_pub = 100, 3
    r1cs, wit = G.square_chain_r1cs(n_con, n_public=n_pub)
    pk, vk = G.setup(r1cs, n_pub, (0x1234567891, 0xabcdef12345, 0x777766665555, 0x3133731337, 0x42424242))
    w = wit([4, 5, 6], 11)
    abc = G.abc_evaluations(r1cs, w, pk["n"])
    want = G.proof_to_uint256x8(G.prove(pk, r1cs, w, 0x9999, 0x7777))
    monkeypatch.setenv("ZKLC_GROTH16_FIXED", "1")
    fixed = Groth16Prover(zctx, pk)
    assert fixed.fixed and fixed.prove(w, abc, 0x9999, 0x7777) == want
    monkeypatch.setenv("ZKLC_GROTH16_FIXED", "0")
    plain = Groth16Prover(zctx, pk)
    assert not plain.fixed and plain.prove(w, abc, 0x9999, 0x7777) == want


def test_proving_key_with_infinity_points_removed(zctx):
    """gnark stores G1.A / G1.B / G2.B without their points at infinity and marks the positions (InfinityA / InfinityB); the prover
    filters the wire values accordingly.  The compacted key + masks must give the proof of the complete key."""
    n_con, n_pub = 40, 2
    r1cs, wit = G.square_chain_r1cs(n_con, n_public=n_pub)
    pk, _ = G.setup(r1cs, n_pub, (0x1234567891, 0xabcdef12345, 0x777766665555, 0x3133731337, 0x42424242))
    inf_a = np.array([p is None for p in pk["A"]])
    inf_b = np.array([p is None for p in pk["B1"]])
    assert inf_a.any() or inf_b.any(), "the test system should have wires that occur in no A or B column"
    assert [p is None for p in pk["B2"]] == list(inf_b)
    compact = dict(pk)
    compact["A"] = [p for p in pk["A"] if p is not None]
    compact["B1"] = [p for p in pk["B1"] if p is not None]
    compact["B2"] = [p for p in pk["B2"] if p is not None]
    compact["infinity_a"], compact["infinity_b"] = inf_a, inf_b
    w = wit([5, 6], 9)
    abc = G.abc_evaluations(r1cs, w, pk["n"])
    full = Groth16Prover(zctx, pk).prove(w, abc, 12345, 67890)
    assert Groth16Prover(zctx, compact).prove(w, abc, 12345, 67890) == full
    assert full == G.proof_to_uint256x8(G.prove(pk, r1cs, w, 12345, 67890))


def _twist_point_outside_g2():
    """a point of E'(Fp2) found by solving for y: with a cofactor of ~2^254 it is (overwhelmingly) not of order r"""
    for x0 in range(1, 50):
        r0, r1 = F._g2_rhs(x0, 1)
        for hint in (False, True):
            try:
                y0, y1 = F._sqrt_fp2(r0, r1, hint)
            except F.ProofInvalid:
                continue
            pt = ((x0, 1), (y0, y1))
            assert B.g2_add(B.g2_mul(G.R - 1, pt), pt) is not None        # [r] pt != O (g2_mul reduces its scalar modulo r)
            return pt
    raise AssertionError("no twist point found")


def test_groth16_verify_on_the_gpu(zctx):
    """`groth16.Verify` (cmd/web-api.go:84) over the MSM + pairing kernels: accepts what the KAT-pinned verifier restatement accepts,
    rejects a changed public input / proof word, and refuses points that are not group elements before any pairing."""
    from zklc_amd.groth16 import Groth16Verifier
    n_pub = 3
    r1cs, wit = G.square_chain_r1cs(60, n_public=n_pub)
    pk, vk = G.setup(r1cs, n_pub, (0x1234567891, 0xabcdef12345, 0x777766665555, 0x3133731337, 0x42424242))
    pubs = [5, 6, 7]
    w = wit(pubs, 11)
    proof = G.proof_to_uint256x8(G.prove(pk, r1cs, w, 0x1234, 0x5678))
    assert G.verify(vk, ((proof[0], proof[1]), ((proof[3], proof[2]), (proof[5], proof[4])), (proof[6], proof[7])), pubs)
    ver = Groth16Verifier(zctx, vk)
    assert ver.verify(proof, pubs) is True
    assert ver.verify(proof, [5, 6, 8]) is False
    assert ver.verify(proof, [5 + G.R, 6, 7]) is True           # public inputs are field elements
    # kSum against the oracle's group arithmetic
    l = vk["K"][0]
    for x, pt in zip(pubs, vk["K"][1:]):
        l = B.add(l, B.mul(x, pt))
    lw, linf = ver.public_input_point(pubs)
    assert not linf and [int(v) for v in lw] == g1_words(l)
    with pytest.raises(ValueError):
        ver.verify(proof, [5, 6])
    # A replaced by another valid G1 point: the equation fails; by a point off the curve: refused
    bad = list(proof)
    two = B.mul(2, B.G1)
    bad[0], bad[1] = two
    assert ver.verify(bad, pubs) is False
    bad[1] = (bad[1] + 1) % F.P_BN254
    with pytest.raises(F.ProofInvalid):
        ver.verify(bad, pubs)
    # B on the twist but outside the r-torsion subgroup: refused by the [r - 1] B = -B test on the G2 MSM kernel
    (x0, x1), (y0, y1) = _twist_point_outside_g2()
    bad = list(proof)
    bad[2], bad[3], bad[4], bad[5] = x1, x0, y1, y0
    with pytest.raises(F.ProofInvalid, match="subgroup"):
        ver.verify(bad, pubs)
    # the point at infinity in the proof: refused
    bad = list(proof)
    bad[6] = bad[7] = 0
    with pytest.raises(F.ProofInvalid):
        ver.verify(bad, pubs)
    # a verifying key with a point outside G2 is refused when the verifier is built
    vk_bad = dict(vk)
    vk_bad["gamma2"] = ((x0, x1), (y0, y1))
    with pytest.raises(F.ProofInvalid):
        Groth16Verifier(zctx, vk_bad)
