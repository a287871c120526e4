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
