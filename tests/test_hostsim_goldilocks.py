"""Goldilocks / Poseidon kernel arithmetic (csrc/*.cuh compiled by g++) against the oracle."""
import ctypes
import random

from oracle import goldilocks as gl
from oracle import poseidon_gl as pg

P = gl.P


def _gl(hostsim):
    hostsim.hostsim_gl_op.restype = ctypes.c_uint64
    hostsim.hostsim_gl_op.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64]
    return hostsim.hostsim_gl_op


def test_field_ops(hostsim):
    op = _gl(hostsim)
    rng = random.Random(1)
    edge = [0, 1, 2, P - 1, P - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**63, P >> 1, 0xFFFFFFFF00000000, 0xFFFFFFFF]
    vals = edge + [rng.randrange(P) for _ in range(200)]
    for a in vals:
        for b in rng.sample(vals, 8) + edge:
            assert op(0, a, b) == (a + b) % P
            assert op(1, a, b) == (a - b) % P
            assert op(2, a, b) == a * b % P
    for _ in range(2000):
        lo, hi = rng.getrandbits(64), rng.getrandbits(64)
        assert op(4, lo, hi) == ((hi << 64) | lo) % P
    for lo, hi in [(2**64 - 1, 2**64 - 1), (0, 2**64 - 1), (2**64 - 1, 0), (0, 0xFFFFFFFF), (0, 0xFFFFFFFF00000000), (P - 1, P - 1)]:
        assert op(4, lo, hi) == ((hi << 64) | lo) % P
    for a in vals[1:30]:
        assert op(3, a, 0) == pow(a, P - 2, P)
    for k in [1, 5, 12, 20, 32]:
        assert op(5, k, 0) == gl.root_of_unity(k)


def test_poseidon_permutation_and_sponge(hostsim):
    rng = random.Random(2)

    def perm(s):
        a = (ctypes.c_uint64 * 12)(*s)
        hostsim.hostsim_poseidon_gl_permute(a)
        return list(a)
    assert perm([0] * 12) == pg._J["kat_permute_zero"]
    assert perm([P - 1] * 12) == pg.permute_naive([P - 1] * 12)
    for _ in range(10):
        s = [rng.randrange(P) for _ in range(12)]
        assert perm(s) == pg.permute_naive(s)
    for n in [0, 1, 3, 4, 5, 8, 9, 16, 17, 135]:
        v = [rng.randrange(P) for _ in range(n)]
        a = (ctypes.c_uint64 * max(n, 1))(*v)
        o = (ctypes.c_uint64 * 4)()
        hostsim.hostsim_poseidon_gl_hash(a, n, o)
        assert list(o) == pg.hash_or_noop(v)
    l, r = [rng.randrange(P) for _ in range(4)], [rng.randrange(P) for _ in range(4)]
    o = (ctypes.c_uint64 * 4)()
    hostsim.hostsim_poseidon_gl_two_to_one((ctypes.c_uint64 * 4)(*l), (ctypes.c_uint64 * 4)(*r), o)
    assert list(o) == pg.two_to_one(l, r)
