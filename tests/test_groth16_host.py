"""Host-side pieces of zklc_amd/groth16.py that need no GPU: the word layouts handed to the kernels (gnark-crypto's Montgomery limbs)
and the point validation `Groth16Verifier` runs before any pairing (gnark: the decoder + `Proof.isValid`, cmd/web-api.go:84)."""
import random

import pytest

from oracle import bn254 as B
from zklc_amd import groth16 as G
from zklc_amd.formats import ProofInvalid


def test_word_layouts_round_trip_and_match_the_oracle():
    rng = random.Random(3)
    for _ in range(20):
        x = rng.randrange(G.P)
        assert G.fp_from_mont_words(G.fp_to_mont_words(x)) == x
    pt = B.mul(12345, B.G1)
    assert G.g1_words(pt) == B.g1_to_words(pt) if hasattr(B, "g1_to_words") else len(G.g1_words(pt)) == 8
    q = B.g2_mul(777, B.G2)
    assert G.g2_words(q) == B.g2_to_words(q)
    assert G.g1_words(None) == [0] * 8 and G.g2_words(None) == [0] * 16
    assert G.fr_to_regular_words(G.R + 5) == [5, 0, 0, 0]
    m = G.fr_to_mont_words(1)
    assert sum(w << (64 * i) for i, w in enumerate(m)) == (1 << 256) % G.R


def test_point_validation_on_the_host():
    G._g1_check(B.G1, "g1")
    G._g1_check(None, "infinity")
    G._g2_check_curve(B.G2, "g2")
    G._g2_check_curve(B.g2_mul(99, B.G2), "g2")
    with pytest.raises(ProofInvalid, match="curve"):
        G._g1_check((1, 3), "p")
    with pytest.raises(ProofInvalid, match="reduced"):
        G._g1_check((G.P + 1, 2), "p")                    # x = 1 + p: the same point with an unreduced coordinate
    (x0, x1), (y0, y1) = B.G2
    with pytest.raises(ProofInvalid, match="curve"):
        G._g2_check_curve(((x0, x1), (y0, (y1 + 1) % G.P)), "q")
    with pytest.raises(ProofInvalid, match="reduced"):
        G._g2_check_curve(((x0 + G.P, x1), (y0, y1)), "q")
    assert G.g2_neg(None) is None
    n = G.g2_neg(B.G2)
    assert n == B.g2_neg(B.G2) and B.g2_add(B.G2, n) is None
