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


def test_loose_arithmetic_edges(hostsim):
    """the non-canonical ("loose") helpers of the Poseidon permutation on their worst-case inputs: values in [p, 2^64)"""
    import numpy as np
    op = _gl(hostsim)
    rng = random.Random(11)
    M = 2**64
    loose = [M - 1, M - 2, P, P + 1, P - 1, 0, 1, 2**32 - 1, 2**32, M - 2**32, M - 2**32 + 1] + [rng.randrange(P, M) for _ in range(50)]
    canon = [0, 1, P - 1, P - 2, 2**32, 2**63] + [rng.randrange(P) for _ in range(50)]
    for a in loose:
        assert op(9, a, 0) == a % P
        for b in canon:
            r = op(7, a, b)
            assert r % P == (a + b) % P and 0 <= r < M
        for b in loose[:12] + canon[:12]:
            r = op(8, a, b)
            assert r % P == a * b % P
    for lo, hi in [(M - 1, M - 1), (0, M - 1), (M - 1, 0), (0, 0xFFFFFFFF), (0, 0xFFFFFFFF00000000), (P - 1, P - 1), (0, 0xFFFFFFFF00000001)] + \
            [(rng.getrandbits(64), rng.getrandbits(64)) for _ in range(2000)]:
        assert op(6, lo, hi) % P == ((hi << 64) | lo) % P
    hostsim.hostsim_gl_acc.restype = ctypes.c_uint64
    hostsim.hostsim_gl_acc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    for n in [1, 2, 12, 13, 200]:
        for vals in ([M - 1] * n, [rng.randrange(P, M) for _ in range(n)], [rng.randrange(M) for _ in range(n)]):
            x = np.array(vals, dtype=np.uint64)
            y = np.array([M - 1 if k % 2 else P - 1 for k in range(n)], dtype=np.uint64)
            want = sum(int(a) * int(b) for a, b in zip(x, y)) % P
            assert hostsim.hostsim_gl_acc(x.ctypes.data, y.ctypes.data, n) == want
    # the carry-free three-column accumulator over 22-bit limb tables (quotient constraints, FRI combine): extreme and random
    # operands, column capacity (511 terms of maximal products without a fold) and the fold-and-restart path
    hostsim.hostsim_gl_acc3.restype = ctypes.c_uint64
    hostsim.hostsim_gl_acc3.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    for n, fold in [(1, 0), (3, 0), (357, 0), (511, 0), (2000, 480), (2000, 384), (1000, 1)]:
        for xs, ks in (([M - 1] * n, [P - 1] * n), ([M - 1] * n, [(1 << 44) - 1 | ((1 << 20) - 1) << 44] * n),
                       ([rng.randrange(M) for _ in range(n)], [rng.randrange(P) for _ in range(n)])):
            x, k = np.array(xs, dtype=np.uint64), np.array(ks, dtype=np.uint64)
            want = sum(int(a) * int(b) for a, b in zip(xs, ks)) % P
            assert hostsim.hostsim_gl_acc3(x.ctypes.data, k.ctypes.data, n, fold) == want, (n, fold)


def test_ntt_group_shift_twiddles_equal_the_definition(hostsim):
    """csrc/goldilocks_ntt_group.cuh: 2^39 is the 64th root of unity of plonky2's generator, x * 2^e, and a butterfly group with
    shift twiddles inside + one table multiplication per element == every butterfly with its full twiddle, for every group size,
    both stage orders and both directions."""
    assert pow(2, 39, P) == pow(1753635133440165772, 2**32 // 64, P)
    assert gl.root_of_unity(6) == pow(2, 39, P)
    rng = random.Random(7)
    f = hostsim.hostsim_gl_mul_2exp
    f.restype = ctypes.c_uint64
    f.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
    edge = [0, 1, P - 1, 2**32 - 1, 2**32, 2**63, 0xFFFFFFFF00000000, P - 2**32]
    for e in range(96):
        for x in edge + [rng.randrange(P) for _ in range(20)]:
            assert f(x, e) == x * pow(2, e, P) % P, (x, e)
    grp = hostsim.hostsim_gl_ntt_group
    grp.restype = ctypes.c_uint32
    grp.argtypes = [ctypes.c_uint32] * 5 + [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    for g in (1, 2, 3, 4):
        for dit in (0, 1):
            for inverse in (0, 1):
                for _ in range(6):
                    logn = rng.randrange(g, 25)
                    s_first = rng.randrange(0, logn - g + 1)
                    J = rng.randrange(1 << (logn - s_first - g))
                    m = 1 << g
                    xs = [rng.choice(edge[:3] + [rng.randrange(P)]) for _ in range(m)]
                    x = (ctypes.c_uint64 * m)(*xs)
                    pl = (ctypes.c_uint64 * m)()
                    assert grp(g, dit, inverse, logn, s_first, J, x, pl) == m
                    assert list(x) == list(pl), (g, dit, inverse, logn, s_first, J)


def test_ntt_group_zero_padded_form_equals_the_general_group(hostsim):
    """the first group of an 8x LDE's first pass (round 6): with the top three index bits of the input known to be zero the group
    evaluates only the butterflies that have a non-zero input, never reads the padded positions, and gives the general group's
    sixteen (eight) values"""
    rng = random.Random(11)
    f = hostsim.hostsim_gl_ntt_group_zero_padded
    f.restype = ctypes.c_uint32
    f.argtypes = [ctypes.c_uint32] * 3 + [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    edge = [0, 1, P - 1, 2**32 - 1, 2**32, P - 2**32]
    for g in (3, 4):
        for _ in range(40):
            logn = rng.randrange(g, 25)
            J = rng.randrange(1 << (logn - g))
            m = 1 << g
            xs = [rng.choice(edge + [rng.randrange(P)] * 3) for _ in range(m)]
            x = (ctypes.c_uint64 * m)(*xs)
            ge = (ctypes.c_uint64 * m)()
            assert f(g, logn, 0, J, x, ge) == m
            assert list(x) == list(ge), (g, logn, J, xs[:m >> 3])


def test_poseidon_asm_generator_selftest_and_committed_file_is_current():
    """csrc/poseidon_gl_asm.inc (the hand-scheduled gfx950 statements of the permutation) is generated by
    tools/gen_poseidon_asm.py: every instruction list is executed by the generator's simulator against big-integer arithmetic
    (x^7, MDS + constants, the 22 fast partial rounds with their in-loop constant prefetch, the initial matrix), the 2-wait-state
    SGPR hazard and the one-SGPR constant-bus rule are checked, and the committed file must be what the generator emits."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("gen_poseidon_asm", os.path.join(ROOT, "tools", "gen_poseidon_asm.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    consts, tables = g.load_constants()
    assert g.selftest(consts, tables)
    # compared in memory: the test never writes into csrc/ (round 5's version regenerated the file in place, which bumped its
    # mtime and made the next incremental build recompile every translation unit)
    text, stats, total = g.render(consts, tables)
    assert open(g.INC_PATH).read() == text, "poseidon_gl_asm.inc is stale: run tools/gen_poseidon_asm.py"
    assert 15000 < total < 17000, total


def test_mul_asm_generator_selftest_and_committed_file_is_current():
    """tools/gen_gl_asm.py: the batched multiplication statements of the NTT passes -- every instruction list is executed against
    big-integer arithmetic (canonical results for any 64-bit operands, edge values included), the flag-hazard and constant-bus rules
    are checked, and csrc/goldilocks_mul_asm.inc is what the generator writes."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_gl_asm", os.path.join(root, "tools", "gen_gl_asm.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.selftest()
    text = open(os.path.join(root, "zk-light-client-implementation_amd", "csrc", "goldilocks_mul_asm.inc")).read()
    assert text == gen.render()
