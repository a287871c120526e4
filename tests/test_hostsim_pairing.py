"""BN254 pairing / Fp2 / G2 device arithmetic (csrc/bn254_pairing.cuh, bn254_fp2.cuh, bn254_ec.cuh compiled by g++)
against the oracle (oracle/bn254_pairing.py, pinned by the reference's Groth16 known-answer proof)."""
import ctypes
import random

import numpy as np
import pytest

from conftest import load_golden
from oracle import bn254 as B
from oracle import bn254_pairing as PR
from test_oracle_pairing import kat_vk


def words32(x):
    m = x * B.MONT_R % B.P
    return [(m >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def from32(w):
    return sum(int(w[i]) << (32 * i) for i in range(8)) * B.inv(B.MONT_R) % B.P


def g1w(p):
    return [0] * 16 if p is None else words32(p[0]) + words32(p[1])


def g2w(q):
    return [0] * 32 if q is None else words32(q[0][0]) + words32(q[0][1]) + words32(q[1][0]) + words32(q[1][1])


def f12w(a):
    return [w for x in PR.f12_flat(a) for w in words32(x)]


def f12_from(w):
    v = [from32(w[8 * i:8 * i + 8]) for i in range(12)]
    c = [(v[2 * i], v[2 * i + 1]) for i in range(6)]
    return ((c[0], c[1], c[2]), (c[3], c[4], c[5]))


def rand_f12(rng):
    return tuple(tuple((rng.randrange(B.P), rng.randrange(B.P)) for _ in range(3)) for _ in range(2))


def test_weak_reduction_edges(hostsim):
    f = hostsim.hostsim_fp_wred
    f.restype = None
    f.argtypes = [ctypes.c_void_p] * 3
    rng = random.Random(1)
    P26 = [(B.P >> (26 * i)) & 0x3ffffff for i in range(10)]
    for trial in range(400):
        k = rng.choice([-64, -63, -17, -2, -1, 0, 1, 2, 9, 31, 63, 64])
        x = rng.randrange(B.P) if trial % 3 else rng.choice([0, 1, B.P - 1])
        target = x + k * B.P if trial % 2 else max(min(x + k * B.P, 64 * B.P), -64 * B.P)
        # spread the value over lazy limbs: random signed limbs (|.| < 2^29) plus a correction in the top limb
        limbs = [rng.randrange(-(1 << 29), 1 << 29) for _ in range(9)]
        rest = target - sum(l << (26 * i) for i, l in enumerate(limbs))
        top, rem = divmod(rest, 1 << 234)
        limbs[0] += rem & 0x3ffffff
        limbs[1] += (rem >> 26) & 0x3ffffff
        val = sum(l << (26 * i) for i, l in enumerate(limbs)) + (top << 234)
        lim = np.array(limbs + [top], dtype=np.int32)
        if abs(top) >= (1 << 30) or abs(val) > 64 * B.P:
            continue
        out_l, out_w = np.zeros(10, dtype=np.int32), np.zeros(8, dtype=np.uint32)
        f(lim.ctypes.data, out_l.ctypes.data, out_w.ctypes.data)
        got = sum(int(l) << (26 * i) for i, l in enumerate(out_l))
        assert (got - val) % B.P == 0 and abs(got) < 2 * B.P, (k, got / B.P)
        assert all(0 <= int(l) < (1 << 26) for l in out_l[:9])
        assert sum(int(w) << (32 * i) for i, w in enumerate(out_w)) == val % B.P


def test_fp12_tower_ops(hostsim):
    f = hostsim.hostsim_f12_op
    f.restype = None
    f.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3
    rng = random.Random(2)

    def call(op, a, b):
        aa, bb, out = np.array(f12w(a), dtype=np.uint32), np.array(f12w(b), dtype=np.uint32), np.zeros(96, dtype=np.uint32)
        f(op, aa.ctypes.data, bb.ctypes.data, out.ctypes.data)
        return f12_from(out)
    edge = (((B.P - 1, B.P - 1),) * 3,) * 2
    for a, b in [(rand_f12(rng), rand_f12(rng)) for _ in range(6)] + [(edge, edge), (PR.F12_ONE, rand_f12(rng))]:
        assert call(0, a, b) == PR.f12_mul(a, b)
        assert call(1, a, b) == PR.f12_sqr(a)
        assert call(2, a, b) == PR.f12_inv(a)
        assert call(3, a, b) == PR.f12_frobenius(a, 1) == PR.f12_pow(a, B.P) if a is edge else call(3, a, b) == PR.f12_frobenius(a, 1)
        assert call(4, a, b) == PR.f12_frobenius(a, 2)
        assert call(5, a, b) == PR.f12_conj(a)
    a = rand_f12(rng)
    assert PR.f12_frobenius(a, 1) == PR.f12_pow(a, B.P) and PR.f12_frobenius(a, 2) == PR.f12_pow(a, B.P * B.P)


def test_g2_curve_ops(hostsim):
    f = hostsim.hostsim_g2_op
    f.restype = ctypes.c_uint32
    f.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3
    p, q = B.g2_mul(1234567, B.G2), B.g2_mul(7654321, B.G2)

    def call(op, a, b):
        aa, bb, out = np.array(g2w(a), dtype=np.uint32), np.array(g2w(b), dtype=np.uint32), np.zeros(32, dtype=np.uint32)
        inf = f(op, aa.ctypes.data, bb.ctypes.data, out.ctypes.data)
        if inf:
            return None
        c = [from32(out[8 * i:8 * i + 8]) for i in range(4)]
        return ((c[0], c[1]), (c[2], c[3]))
    assert call(0, p, q) == B.g2_add(p, q) and call(1, p, q) == B.g2_add(p, B.g2_neg(q)) and call(2, p, q) == B.g2_add(p, p)
    assert call(0, p, p) == B.g2_add(p, p) and call(1, p, p) is None
    assert call(3, p, q) == B.g2_add(B.g2_add(p, q), p)


def _pairing(hostsim, pairs, stage=1):
    f = hostsim.hostsim_pairing
    f.restype = ctypes.c_uint32
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    a = np.array([w for p, _ in pairs for w in g1w(p)], dtype=np.uint32)
    b = np.array([w for _, q in pairs for w in g2w(q)], dtype=np.uint32)
    out = np.zeros(96, dtype=np.uint32)
    one = f(a.ctypes.data, b.ctypes.data, len(pairs), stage, out.ctypes.data)
    return bool(one), f12_from(out)


def test_pairing_value_matches_oracle(hostsim):
    p, q = B.mul(0xABCDEF, B.G1), B.g2_mul(0x13579B, B.G2)
    one, gt = _pairing(hostsim, [(p, q)])
    assert not one and gt == PR.pairing(p, q)
    one, gt = _pairing(hostsim, [(p, q), (B.neg(p), q)])
    assert one and gt == PR.F12_ONE
    one, gt = _pairing(hostsim, [(None, q), (p, None)])
    assert one


def test_reference_groth16_kat_through_the_device_pairing_code(hostsim):
    j = load_golden("groth16_kat.json")
    vk = kat_vk(j)
    proof, inputs = [int(x) for x in j["proof"]], [int(x) for x in j["inputs"]]

    def pairs(proof, inputs):
        a = (proof[0], proof[1])
        b = ((proof[3], proof[2]), (proof[5], proof[4]))
        c = (proof[6], proof[7])
        l = vk["ic"][0]
        for s, pt in zip(inputs, vk["ic"][1:]):
            l = B.add(l, B.mul(s, pt))
        return [(a, b), (c, vk["delta_neg"]), (vk["alpha"], vk["beta_neg"]), (l, vk["gamma_neg"])]
    assert _pairing(hostsim, pairs(proof, inputs))[0]
    assert not _pairing(hostsim, pairs(proof, [int(x) for x in j["incorrect_inputs"]]))[0]
