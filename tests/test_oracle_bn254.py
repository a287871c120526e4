"""Pins the BN254 oracles: Python group law vs the curve equation / generator, C port vs Python."""
import random

import numpy as np

from oracle import bn254 as bn
from oracle import cport


def words(pt):
    return bn.to_mont_words(pt[0]) + bn.to_mont_words(pt[1])


def unwords(w):
    return (bn.from_mont_words(w[:4]), bn.from_mont_words(w[4:]))


def scalar_words(s):
    return [(s >> (64 * i)) & (2**64 - 1) for i in range(4)]


def test_group_law_basics():
    G = bn.G1
    assert bn.is_on_curve(G) and bn.mul(bn.R, G) is None
    a, b = bn.mul(5, G), bn.mul(7, G)
    assert bn.add(a, b) == bn.mul(12, G) and bn.is_on_curve(bn.add(a, b))
    assert bn.add(a, bn.neg(a)) is None


def test_c_point_generator_and_msm_match_python():
    rng = random.Random(1)
    n = 24
    pts = cport.bn254_gen_points(n, 7, 11)
    py_pts = [bn.mul(7 + 11 * i, bn.G1) for i in range(n)]
    assert [unwords([int(x) for x in pts[i]]) for i in range(n)] == py_pts
    scalars = [rng.randrange(bn.R) for _ in range(n)]
    scalars[0], scalars[1], scalars[2] = 0, 1, bn.R - 1
    sc = np.array([scalar_words(s) for s in scalars], dtype=np.uint64)
    want = bn.msm(scalars, py_pts)
    for naive in (True, False):
        out, inf, _ = cport.bn254_msm(pts, sc, naive=naive)
        assert not inf and unwords([int(x) for x in out]) == want


def test_c_msm_bucket_vs_naive_medium():
    rng = np.random.default_rng(2)
    n = 300
    pts = cport.bn254_gen_points(n, 3, 5)
    sc = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    sc[:, 3] >>= np.uint64(3)  # < 2^252 < r
    a, ia, _ = cport.bn254_msm(pts, sc, naive=True)
    b, ib, _ = cport.bn254_msm(pts, sc, nthreads=2)
    assert ia == ib is False and np.array_equal(a, b)
    # cancelling pair -> infinity
    two = np.vstack([pts[:1], pts[:1]])
    s2 = np.array([scalar_words(5), scalar_words(bn.R - 5)], dtype=np.uint64)
    _, inf, _ = cport.bn254_msm(two, s2)
    assert inf


def test_fr_ntt_oracle_is_the_dft():
    import random
    from oracle import bn254_fr as FR
    rng = random.Random(3)
    for log_n in [0, 1, 3, 5]:
        a = [rng.randrange(FR.R) for _ in range(1 << log_n)]
        assert FR.ntt(a) == FR.naive_dft(a)
        assert FR.ntt(a, coset=True) == FR.naive_dft(a, FR.GENERATOR)
        assert FR.ntt(FR.ntt(a), inverse=True) == a
        assert FR.ntt(FR.ntt(a, coset=True), inverse=True, coset=True) == a
    for x in [0, 1, FR.R - 1, rng.randrange(FR.R)]:
        assert FR.from_mont_words(FR.to_mont_words(x)) == x


def test_g2_group_law_basics():
    B = bn
    assert B.g2_is_on_curve(B.G2) and B.g2_mul(B.R, B.G2) is None
    a, b = B.g2_mul(5, B.G2), B.g2_mul(7, B.G2)
    assert B.g2_add(a, b) == B.g2_mul(12, B.G2) and B.g2_add(a, B.g2_neg(a)) is None
    assert B.g2_from_words(B.g2_to_words(a)) == a
    assert B.g2_msm([3, 4], [a, b]) == B.g2_mul(15 + 28, B.G2)
