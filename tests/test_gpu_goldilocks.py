"""GPU parity: Goldilocks NTT/LDE, Poseidon and Merkle kernels through the C ABI vs the oracle."""
import numpy as np
import pytest

from oracle import cport
from oracle import goldilocks as gl
from oracle import poseidon_gl as pg

pytestmark = pytest.mark.gpu
P = gl.P
INV, IN_BR, OUT_BR = 1, 2, 4


def rand_gl(rng, shape):
    # uniform in [0, p): rejection is irrelevant at 2^-32, clamp instead
    a = rng.integers(0, 2**64, size=shape, dtype=np.uint64)
    return np.where(a >= np.uint64(P), a - np.uint64(P), a)


def bitrev_perm(n):
    bits = n.bit_length() - 1
    return np.array([int(format(i, "0%db" % bits)[::-1], 2) if bits else 0 for i in range(n)])


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 8, 11, 12, 13, 14, 16, 18])
def test_ntt_forward_matches_oracle(zctx, log_n):
    rng = np.random.default_rng(log_n)
    batch = 3 if log_n < 16 else 2
    a = rand_gl(rng, (batch, 1 << log_n))
    want = cport.gl_ntt(a, nthreads=4)
    got = zctx.gl_ntt(a)
    assert np.array_equal(got, want)
    # bit-reversed output flag = same values, permuted
    got_br = zctx.gl_ntt(a, flags=OUT_BR)
    assert np.array_equal(got_br, want[:, bitrev_perm(1 << log_n)])
    # bit-reversed input (DIT path)
    got_dit = zctx.gl_ntt(a[:, bitrev_perm(1 << log_n)], flags=IN_BR)
    assert np.array_equal(got_dit, want)


@pytest.mark.parametrize("log_n", [1, 7, 12, 13, 17])
def test_ntt_inverse_roundtrip(zctx, log_n):
    rng = np.random.default_rng(100 + log_n)
    a = rand_gl(rng, (2, 1 << log_n))
    f = zctx.gl_ntt(a)
    assert np.array_equal(zctx.gl_ntt(f, flags=INV), a)
    assert np.array_equal(zctx.gl_ntt(a, flags=INV), cport.gl_ntt(a, inverse=True))
    # DIF forward (bit-reversed out) then DIT inverse (bit-reversed in): no permutation pass at all
    assert np.array_equal(zctx.gl_ntt(zctx.gl_ntt(a, flags=OUT_BR), flags=INV | IN_BR), a)


def test_small_ntt_matches_python_definition(zctx):
    rng = np.random.default_rng(9)
    a = rand_gl(rng, (1, 32))
    assert [int(x) for x in zctx.gl_ntt(a)[0]] == gl.naive_dft([int(x) for x in a[0]])


@pytest.mark.parametrize("log_n,rate_bits,batch", [(3, 3, 2), (9, 3, 5), (12, 3, 3), (10, 1, 2), (14, 3, 2), (12, 0, 1)])
def test_lde_matches_oracle(zctx, log_n, rate_bits, batch):
    rng = np.random.default_rng(log_n * 10 + rate_bits)
    c = rand_gl(rng, (batch, 1 << log_n))
    want = cport.gl_lde(c, rate_bits, 7, nthreads=4)
    assert np.array_equal(zctx.gl_lde(c, rate_bits, 7), want)
    N = 1 << (log_n + rate_bits)
    assert np.array_equal(zctx.gl_lde(c, rate_bits, 7, flags=OUT_BR), want[:, bitrev_perm(N)])
    # evaluation property: value k is the polynomial at 7 * w^k
    w = gl.root_of_unity(log_n + rate_bits)
    for k in [0, 1, N - 1]:
        assert int(want[0, k]) == gl.eval_poly([int(x) for x in c[0]], 7 * pow(w, k, P) % P)


@pytest.mark.parametrize("log_n,rate_bits", [(18, 3), (19, 3), (13, 0)])
def test_big_tile_sizes_match_oracle(zctx, log_n, rate_bits):
    """2^21 (the Ed25519 circuit's LDE) and 2^22 take the 2^13-element LDS tile (two passes instead of three), 2^13 a single pass:
    forward / inverse / DIT transforms and the LDE against the C oracle on one polynomial, round trip on a second"""
    rng = np.random.default_rng(500 + log_n)
    logN = log_n + rate_bits
    a = rand_gl(rng, (2, 1 << logN))
    want = cport.gl_ntt(a[:1], nthreads=8)
    f = zctx.gl_ntt(a, flags=OUT_BR)
    br = bitrev_fast(logN)
    assert np.array_equal(f[:1][:, br], want)
    assert np.array_equal(zctx.gl_ntt(f, flags=INV | IN_BR), a)
    assert np.array_equal(zctx.gl_ntt(a[:1], flags=INV), cport.gl_ntt(a[:1], inverse=True, nthreads=8))
    if rate_bits:
        c = rand_gl(rng, (1, 1 << log_n))
        assert np.array_equal(zctx.gl_lde(c, rate_bits, 7), cport.gl_lde(c, rate_bits, 7, nthreads=8))


def bitrev_fast(bits):
    idx = np.arange(1 << bits, dtype=np.uint64)
    out = np.zeros_like(idx)
    for b in range(bits):
        out |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(bits - 1 - b)
    return out.astype(np.int64)


def test_lde_c3_shape_linearity(zctx):
    """Full C3 shape (2^17 -> 2^20) through size-independent properties: linearity and
    agreement with the oracle on one polynomial."""
    rng = np.random.default_rng(77)
    a, b = rand_gl(rng, (1, 1 << 17)), rand_gl(rng, (1, 1 << 17))
    s = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
    la, lb, ls = (zctx.gl_lde(x, 3, 7, flags=OUT_BR) for x in (a, b, s))
    assert np.array_equal(((la.astype(object) + lb.astype(object)) % P).astype(np.uint64), ls)
    assert np.array_equal(la[:, bitrev_perm(1 << 20)], cport.gl_lde(a, 3, 7, nthreads=8))


def test_poseidon_permute(zctx):
    rng = np.random.default_rng(3)
    st = rand_gl(rng, (300, 12))
    st[0] = 0
    st[1] = P - 1
    out = zctx.poseidon_gl_permute(st)
    assert [int(x) for x in out[0]] == pg._J["kat_permute_zero"]  # goldilocks_test.go:47-53
    for i in range(300):
        assert [int(x) for x in out[i]] == cport.poseidon_gl_permute([int(x) for x in st[i]])


@pytest.mark.parametrize("width,log_leaves,cap", [(1, 0, 0), (3, 3, 0), (4, 4, 4), (5, 6, 2), (135, 9, 4), (20, 12, 4), (234, 10, 4), (16, 1, 0)])
def test_merkle_commit(zctx, width, log_leaves, cap):
    rng = np.random.default_rng(width * 31 + log_leaves)
    mat = rand_gl(rng, (width, 1 << log_leaves))
    cap_gpu, levels = zctx.gl_merkle_commit(mat, cap)
    want = cport.gl_merkle_commit(mat, cap, nthreads=4)
    assert len(levels) == len(want)
    for g, w in zip(levels, want):
        assert np.array_equal(g, w)
    if log_leaves <= 6:  # pin the C port against the Python restatement too, and open every leaf
        leaves = [[int(mat[p, i]) for p in range(width)] for i in range(1 << log_leaves)]
        cap_py, layers = pg.merkle_tree(leaves, cap)
        assert [[int(x) for x in d] for d in cap_gpu] == cap_py
        for i in range(1 << log_leaves):
            sib = [[int(x) for x in levels[l][(i >> l) ^ 1]] for l in range(log_leaves - cap)]
            assert pg.merkle_verify(leaves[i], i, sib, cap_py)


def test_hash_no_pad_kat_through_merkle(zctx):
    # tests/public_inputs_hash_test.go:54-55 is hash_no_pad([0,1,x]); a width-3 leaf is <= 4 elements (noop),
    # so pin the sponge through a width-5 leaf against the Python oracle and the KAT through the permutation
    kat = pg._J["kat_hash_no_pad"]
    st = np.zeros((1, 12), dtype=np.uint64)
    st[0, :3] = kat["in"]
    assert [int(x) for x in zctx.poseidon_gl_permute(st)[0, :4]] == kat["out"]
