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


def test_poseidon_bn254(hostsim):
    """crypto/plonky2_bn128/src/poseidon_bn128.rs:133-180 KATs + hasher packing (config.rs:132-199)."""
    from oracle import poseidon_bn254 as pb
    rng = random.Random(1)
    w8l = lambda x: [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]

    def perm(st):
        a = (ctypes.c_uint32 * 32)(*sum([w8l(x) for x in st], []))
        hostsim.hostsim_poseidon_bn254_permute(a)
        return [sum(a[8 * i + k] << (32 * k) for k in range(8)) for i in range(4)]
    def perm_coop(st):       # the four-lane form used for the small trees (csrc/poseidon_bn254.cuh), walked lane by lane
        a = (ctypes.c_uint32 * 32)(*sum([w8l(x) for x in st], []))
        hostsim.hostsim_poseidon_bn254_permute_coop(a)
        return [sum(a[8 * i + k] << (32 * k) for k in range(8)) for i in range(4)]
    for k in pb.KATS:
        assert perm(k["in"]) == k["out"]
        assert perm_coop(k["in"]) == k["out"]
    for _ in range(3):
        s = [rng.randrange(pb.R) for _ in range(4)]
        assert perm(s) == pb.permute(s)
        assert perm_coop(s) == pb.permute(s)
    GP = 2**64 - 2**32 + 1
    for n in [0, 1, 3, 4, 8, 9, 10, 18, 19, 135]:
        v = [rng.randrange(GP) for _ in range(n)]
        a = (ctypes.c_uint64 * max(1, n))(*v)
        o = (ctypes.c_uint32 * 8)()
        hostsim.hostsim_poseidon_bn254_hash(a, n, o)
        assert sum(o[k] << (32 * k) for k in range(8)) == pb.hash_or_noop(v)
    l, r = rng.randrange(pb.R), rng.randrange(pb.R)
    o = (ctypes.c_uint32 * 8)()
    hostsim.hostsim_poseidon_bn254_two_to_one((ctypes.c_uint32 * 8)(*w8l(l)), (ctypes.c_uint32 * 8)(*w8l(r)), o)
    assert sum(o[k] << (32 * k) for k in range(8)) == pb.two_to_one(l, r)
