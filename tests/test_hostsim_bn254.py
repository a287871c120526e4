"""BN254 Fp / G1 kernel arithmetic (csrc/bn254_*.cuh compiled by g++) against the oracle."""
import ctypes
import random

from oracle import bn254 as bn

P = bn.P


def w8(x):
    ws = bn.to_mont_words(x)
    return (ctypes.c_uint32 * 8)(*[(ws[i // 2] >> (32 * (i % 2))) & 0xFFFFFFFF for i in range(8)])


def unw(o):
    return bn.from_mont_words([o[2 * i] | (o[2 * i + 1] << 32) for i in range(4)])


def test_fp_ops(hostsim):
    rng = random.Random(4)

    def fop(op, a, b=0):
        o = (ctypes.c_uint32 * 8)()
        hostsim.hostsim_fp_op(op, w8(a), w8(b), o)
        return unw(o)
    edge = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 3]
    vals = edge + [rng.randrange(P) for _ in range(100)]
    for a in vals:
        for b in rng.sample(vals, 6) + edge:
            assert fop(0, a, b) == (a + b) % P
            assert fop(1, a, b) == (a - b) % P
            assert fop(2, a, b) == a * b % P
            assert fop(5, a, b) == ((a * b - a - 2 * b) * (a + b + a * a)) % P
        assert fop(3, a) == a * a % P
    for a in vals[1:10]:
        assert fop(4, a) == pow(a, P - 2, P)


def test_g1_ops(hostsim):
    rng = random.Random(5)

    def pt16(pt):
        if pt is None:
            return (ctypes.c_uint32 * 16)(), 1
        return (ctypes.c_uint32 * 16)(*(list(w8(pt[0])) + list(w8(pt[1])))), 0

    def gop(op, p, q, n=0):
        a, ai = pt16(p)
        b, bi = pt16(q)
        o = (ctypes.c_uint32 * 16)()
        inf = hostsim.hostsim_g1_op(op, a, ai, b, bi, n, o)
        return None if inf else (unw(o[0:8]), unw(o[8:16]))
    G = bn.G1
    A, B = bn.mul(12345, G), bn.mul(99999, G)
    assert gop(0, A, B) == bn.add(A, B)
    assert gop(1, A, B) == bn.add(A, bn.neg(B))
    assert gop(2, A, None) == bn.add(A, A)
    assert gop(0, A, A) == bn.add(A, A)        # mixed add hitting the doubling case
    assert gop(1, A, A) is None                # P - P
    assert gop(0, None, B) == B and gop(0, A, None) == A
    assert gop(3, A, B) == bn.mul(2, bn.add(A, B))  # general add, doubling case
    assert gop(5, A, B) == bn.add(A, B)
    assert gop(5, A, bn.neg(A)) is None
    assert gop(5, None, B) == B
    assert gop(4, A, B, 50) == bn.add(A, bn.mul(50, B))
    assert gop(4, None, G, 7) == bn.mul(7, G)  # infinity + G, G + G (doubling), ...
    for _ in range(10):
        X, Y = bn.hash_to_curve(7, rng.randrange(10**6)), bn.hash_to_curve(7, rng.randrange(10**6))
        assert bn.is_on_curve(X) and gop(0, X, Y) == bn.add(X, Y)
