"""The exact kernel arithmetic (csrc/*.cuh compiled by g++) against the oracle."""
import ctypes
import hashlib
import os
import random

import pytest

from conftest import load_golden, near_sets, near_set_arrays
from edcases import edge_cases, synthetic_set
from oracle import ed25519_ref as ref

P = ref.P


def _w(x, n=8):
    return (ctypes.c_uint32 * n)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)])


def _fe_op(lib, op, a, b=0):
    out = (ctypes.c_uint32 * 8)()
    lib.hostsim_fe_op(op, _w(a), _w(b), out)
    return sum(out[i] << (32 * i) for i in range(8))


def test_field_ops(hostsim):
    rng = random.Random(1)
    M = 2**255
    edge = [0, 1, 2, 19, 38, P - 1, P, P + 1, M - 1, M - 19, M - 20, 2**254, 2**26 - 1, 2**51 - 1, (1 << 230) - 1]
    vals = edge + [rng.getrandbits(255) for _ in range(150)]
    for a in vals:
        for b in rng.sample(vals, 8) + edge[:6] + [M - 1]:
            assert _fe_op(hostsim, 0, a, b) == (a + b) % P
            assert _fe_op(hostsim, 1, a, b) == (a - b) % P
            assert _fe_op(hostsim, 2, a, b) == (a * b) % P
            assert _fe_op(hostsim, 8, a, b) == ((2 * a + b) * (b - 2 * a)) % P
        assert _fe_op(hostsim, 3, a) == a * a % P
        assert _fe_op(hostsim, 7, a) == 2 * a * a % P
        assert _fe_op(hostsim, 6, a) == a % P
    for a in vals[:40]:
        assert _fe_op(hostsim, 4, a) == pow(a, P - 2, P)
        assert _fe_op(hostsim, 5, a) == pow(a, (P - 5) // 8, P)


def test_scalar_reduce(hostsim):
    rng = random.Random(2)
    L = ref.L
    xs = [0, 1, L - 1, L, L + 1, 2**512 - 1, 2**512 - L, (2**512 // L) * L, (2**512 // L) * L - 1, 2**252, 2**253 - 1, 2**504]
    xs += [rng.getrandbits(512) for _ in range(300)]
    for x in xs:
        out = (ctypes.c_uint32 * 8)()
        hostsim.hostsim_sc_reduce512(_w(x, 16), out)
        assert sum(out[i] << (32 * i) for i in range(8)) == x % L
    for x in [0, 1, L - 1, L, L + 1, 2**256 - 1, 2**252, 2**253]:
        assert hostsim.hostsim_sc_is_canonical(_w(x)) == int(x < L)


def test_sha512(hostsim):
    for n in [0, 1, 3, 55, 111, 112, 113, 127, 128, 129, 200, 239, 240, 241, 255, 256, 257, 1000]:
        m = os.urandom(n)
        o = ctypes.create_string_buffer(64)
        hostsim.hostsim_sha512(m, n, o)
        assert o.raw == hashlib.sha512(m).digest()


def test_base_table(hostsim):
    T = (ctypes.c_uint32 * (128 * 30))()
    hostsim.hostsim_base_table(T)
    def fe(i):  # ten radix-2^25.5 limbs
        v, pos = 0, 0
        for k in range(10):
            v += T[i + k] << pos
            pos += 26 if k % 2 == 0 else 25
        return v
    for j in [1, 2, 3, 7, 64, 127, 128]:
        x, y = ref.pt_affine(ref.pt_mul(j, ref.BASE))
        o = (j - 1) * 30
        assert (fe(o), fe(o + 10), fe(o + 20)) == ((y + x) % P, (y - x) % P, 2 * ref.D * x * y % P)


@pytest.mark.parametrize("name", near_sets())
def test_near_fixtures(hostsim, name):
    j = load_golden(name)
    msg, approvals, validators = near_set_arrays(j)
    ok = sum(hostsim.hostsim_ed25519_verify(va[-48:-16], ap[2:], msg, len(msg))
             for ap, va in zip(approvals, validators) if len(ap) == 66)
    assert ok == j["expect_valid"]


def test_edge_cases_match_oracle(hostsim):
    for pk, sig, msg, label in edge_cases():
        assert hostsim.hostsim_ed25519_verify(pk, sig, msg, len(msg)) == int(ref.verify(pk, sig, msg)), label


def test_synthetic_with_corruption(hostsim):
    pks, sigs, msg = synthetic_set(96, seed=3, corrupt_every=7)
    for i, (pk, sg) in enumerate(zip(pks, sigs)):
        exp = int(i % 7 != 6)
        assert hostsim.hostsim_ed25519_verify(pk, sg, msg, len(msg)) == exp
        assert int(ref.verify(pk, sg, msg)) == exp
