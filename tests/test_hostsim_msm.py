"""The BN254 multi-scalar multiplication of csrc/bn254_msm.hip, walked lane by lane on the CPU (tests/hostsim: the same
bn254_msm_lane.cuh functions, plan, digit codes, tile counting sort, pipelined bucket loop, segment sums) against the oracle's
naive multi-exponentiation (oracle/c/bn254_oracle.c) -- SURVEY 8(d) C4 scalar classes: uniform, witness-like, adversarial."""
import ctypes

import numpy as np
import pytest

from oracle import cport, bn254 as B

R = B.R


def _scalars(kind, n, rng):
    if kind == "uniform":
        vals = [int(rng.integers(0, 2**63)) * (2**192) % R + int(rng.integers(0, 2**63)) * int(rng.integers(0, 2**63)) for _ in range(n)]
        vals = [v % R for v in vals]
    elif kind == "witness":          # 50 % in {0, 1}, 30 % < 2^64, 20 % uniform
        vals = []
        for _ in range(n):
            u = rng.random()
            if u < 0.5:
                vals.append(int(rng.integers(0, 2)))
            elif u < 0.8:
                vals.append(int(rng.integers(0, 2**63)))
            else:
                vals.append((int(rng.integers(0, 2**63)) << 190 | int(rng.integers(0, 2**63))) % R)
    elif kind == "equal":            # adversarial: every scalar the same -> one bucket per window takes everything
        vals = [0x1234567890ABCDEF1122334455667788990011223344556677889900AABBCC % R] * n
    elif kind == "top":              # largest digits: r - 1, 2^15 multiples (the +2^15 digit code), unreduced scalars (>= r)
        vals = [R - 1, 1 << 15, (1 << 15) + (1 << 31), (1 << 253) + 12345, 2**256 - 1, R, R + 7] * (n // 7 + 1)
        vals = vals[:n]
    else:
        raise ValueError(kind)
    return np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for v in vals], dtype=np.uint64)


def _run(hostsim, pts, sc, mode=0, heavy_min=1 << 30, heavy_threads=4):
    n = pts.shape[0]
    out = (ctypes.c_uint32 * 16)()
    inf = hostsim.hostsim_msm_g1(pts.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p), n, mode, heavy_min, heavy_threads, out)
    got = np.array([out[2 * i] | (out[2 * i + 1] << 32) for i in range(8)], dtype=np.uint64)
    return got, bool(inf)


@pytest.mark.parametrize("kind,n", [("uniform", 300), ("witness", 300), ("equal", 200), ("top", 70), ("uniform", 2500), ("uniform", 1),
                                    ("equal", 3000), ("witness", 5000)])
def test_msm_lane_walk_matches_oracle(hostsim, kind, n):
    rng = np.random.default_rng(5 + n)
    pts = cport.bn254_gen_points(n, 5, 3)
    if kind == "witness":
        pts[3] = 0          # points at infinity (all-zero encoding) take no part, whatever their scalar
        pts[n - 1] = 0
    sc = _scalars(kind, n, rng)
    live = [i for i in range(n) if pts[i].any()]
    want, winf, _ = cport.bn254_msm(pts[live], _reduced(sc[live]), naive=True)
    got, inf = _run(hostsim, pts, sc)
    assert inf == winf and (inf or np.array_equal(got, want))
    # the workgroup form of the combine (every bucket cut into >= 2 slices, 3 strided partial sums) gives the same point
    got2, inf2 = _run(hostsim, pts, sc, heavy_min=1, heavy_threads=3)
    assert inf2 == inf and np.array_equal(got2, got)
    # and so does the one-lane-per-bucket walk on the raw points
    got3, inf3 = _run(hostsim, pts, sc, mode=1, heavy_min=2, heavy_threads=3)
    assert inf3 == inf and np.array_equal(got3, got)
    # and the slices over records of ten 32-bit limbs per coordinate (ZKLC_MSM_PACKED=0; mode 0 reads the packed 64-byte records)
    got4, inf4 = _run(hostsim, pts, sc, mode=2)
    assert inf4 == inf and np.array_equal(got4, got)


def _reduced(sc):
    """the oracle's naive MSM takes reduced scalars: reduce mod r on the host (the product reduces unreduced scalars itself)"""
    out = np.zeros_like(sc)
    for i, row in enumerate(sc):
        v = sum(int(row[k]) << (64 * k) for k in range(4)) % R
        out[i] = [(v >> (64 * k)) & (2**64 - 1) for k in range(4)]
    return out


def test_all_points_equal_and_cancelling(hostsim):
    """all points equal (every bucket addition after the first is a doubling) and P, -P pairs (sums hit infinity)"""
    n = 64
    one = cport.bn254_gen_points(1, 9, 1)
    pts = np.repeat(one, n, axis=0)
    rng = np.random.default_rng(3)
    sc = _scalars("uniform", n, rng)
    sc[: n // 2] = sc[0]                      # same scalar, same point: P = Q in the mixed addition
    want, winf, _ = cport.bn254_msm(pts, sc, naive=True)
    got, inf = _run(hostsim, pts, sc)
    assert not inf and np.array_equal(got, want)
    # s * P + (r - s) * P = infinity
    sc2 = sc.copy()
    for i in range(0, n, 2):
        v = sum(int(sc[i][k]) << (64 * k) for k in range(4))
        w = (R - v) % R
        sc2[i + 1] = [(w >> (64 * k)) & (2**64 - 1) for k in range(4)]
    got, inf = _run(hostsim, pts, sc2)
    assert inf


def test_plan_and_digit_codes(hostsim):
    out = (ctypes.c_uint32 * 8)()
    for n in (1, 63, 64, 2047, 2048, 1 << 15, (1 << 19) - 1, 1 << 19, 1 << 22, (1 << 22) + 5):
        hostsim.hostsim_msm_plan(ctypes.c_uint64(n), out)
        n_pad, c, windows, bpw, total, chunks, chunk_len, seg = list(out)
        assert n_pad % 8 == 0 and n <= n_pad < n + 8 and c <= 16 and windows == 254 // c + 1 and bpw == 1 << (c - 1)
        assert total == windows * bpw and 1 <= chunks <= 16 and chunk_len % 8 == 0 and chunks * chunk_len >= n_pad
        assert bpw * 4 <= 160 * 1024          # a window's counters fit the LDS of one workgroup
    # digits reassemble the scalar; codes decode to (bucket, sign)
    rng = np.random.default_rng(11)
    digs, codes = (ctypes.c_int * 64)(), (ctypes.c_uint32 * 64)()
    for c in (4, 8, 11, 14, 16):
        for _ in range(50):
            v = int(rng.integers(0, 2**63)) << 191 | int(rng.integers(0, 2**63)) << 100 | int(rng.integers(0, 2**63))
            v %= R
            if _ % 10 == 0:
                v = R - 1 - _
            s4 = (ctypes.c_uint64 * 4)(*[(v >> (64 * k)) & (2**64 - 1) for k in range(4)])
            carry = hostsim.hostsim_msm_digit_roundtrip(s4, c, digs, codes)
            windows = 254 // c + 1
            assert carry == 0
            assert sum(digs[w] << (c * w) for w in range(windows)) == v
            for w in range(windows):
                d = digs[w]
                assert -(2 ** (c - 1)) < d <= 2 ** (c - 1)
                code = codes[w]
                assert (code == 0) == (d == 0) and code < 65536
                if d:
                    neg = code > 0x8000
                    mag = 0x10000 - code if neg else code
                    assert (neg, mag) == (d < 0, abs(d))


def test_g2_lane_walk_with_quad_doublings_matches_python(hostsim):
    """G2 (coordinates in Fp2) through the same lane code, the final doublings by a quad of lanes (ecq_stage* over Fp2Field, as
    msm_final_kernel runs them since round 4) -- against the oracle's Python G2 arithmetic; scalars with digits in every window"""
    n = 12
    rng = np.random.default_rng(2)
    cur, step, pts, plist = B.g2_mul(31337, B.G2), B.g2_mul(99, B.G2), [], []
    for _ in range(n):
        pts.append(B.g2_to_words(cur))
        plist.append(cur)
        cur = B.g2_add(cur, step)
    pts = np.array(pts, dtype=np.uint64)
    vals = [(int(rng.integers(0, 2**63)) << 190 | int(rng.integers(0, 2**63)) << 100 | int(rng.integers(0, 2**63))) % R for _ in range(n - 2)] + [R - 1, 1]
    sc = np.array([[(v >> (64 * k)) & (2**64 - 1) for k in range(4)] for v in vals], dtype=np.uint64)
    out = (ctypes.c_uint32 * 32)()
    outs = []
    for mode in (0, 1, 2):       # 0: packed records + quad doublings (the product path); 2: unpacked records; 1: one lane per bucket, ec_double
        inf = hostsim.hostsim_msm_g2(pts.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p), n, mode, 1 << 30, 4, out)
        assert not inf
        outs.append([out[2 * i] | (out[2 * i + 1] << 32) for i in range(16)])
    want = B.g2_to_words(B.g2_msm(vals, plist))
    assert outs[0] == outs[1] == outs[2] == [int(x) for x in want]


def test_glv_split_and_the_split_msm(hostsim):
    """the endomorphism split of the G1 multi-exponentiation (bn254_msm_lane.cuh: msm_glv_split): k = k1 + k2 lambda (mod r) with
    |k1|, |k2| < 2^127 for random and extreme scalars; lambda P = (beta x, y) on the oracle's curve arithmetic; and the whole lane walk
    WITH the split (>= 256 points: 2 n items, 8 windows at c = 16 -- here smaller windows, same code) equals the oracle's naive sum"""
    lam = 4407920970296243842393367215006156084916469457145843978461
    beta = 2203960485148121921418603742825762020974279258880205651966
    assert (lam * lam + lam + 1) % R == 0 and pow(beta, 3, B.P) == 1
    assert B.mul(lam, B.G1) == (beta * B.G1[0] % B.P, B.G1[1])
    rng = np.random.default_rng(11)
    f = hostsim.hostsim_msm_glv_split
    f.restype = None
    cases = [0, 1, 2, R - 1, R - 2, lam, R - lam, 1 << 253, (1 << 127), (1 << 127) - 1, 2**256 - 1] + \
        [int(rng.integers(0, 2**63)) << 191 | int(rng.integers(0, 2**63)) << 128 | int(rng.integers(0, 2**63)) << 64 | int(rng.integers(0, 2**63))
         for _ in range(3000)]
    for k in cases:
        sc = np.array([(k >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
        m1, m2, negs = (ctypes.c_uint32 * 4)(), (ctypes.c_uint32 * 4)(), (ctypes.c_uint32 * 2)()
        f(sc.ctypes.data_as(ctypes.c_void_p), m1, m2, negs)
        k1 = sum(int(m1[i]) << (32 * i) for i in range(4)) * (-1 if negs[0] else 1)
        k2 = sum(int(m2[i]) << (32 * i) for i in range(4)) * (-1 if negs[1] else 1)
        assert abs(k1) < 1 << 127 and abs(k2) < 1 << 127 and (k1 + k2 * lam - k) % R == 0, k
    # the full walk with the split: 300 and 2500 points are in the parametrised test above (mode 0 = the product plan); here the
    # plan itself and a case with points at infinity and unreduced scalars
    plan = (ctypes.c_uint32 * 8)()
    n = 700
    pts = cport.bn254_gen_points(n, 13, 5)
    pts[5] = 0
    pts[n - 2] = 0
    sc = _scalars("top", n, rng)
    live = [i for i in range(n) if pts[i].any()]
    want, winf, _ = cport.bn254_msm(pts[live], _reduced(sc[live]), naive=True)
    got, inf = _run(hostsim, pts, sc)
    assert inf == winf and np.array_equal(got, want)
    got2, inf2 = _run(hostsim, pts, sc, mode=2)          # no split, unpacked records: the same point
    assert inf2 == inf and np.array_equal(got2, got)


@pytest.mark.parametrize("kind,n", [("uniform", 300), ("witness", 500), ("equal", 200), ("top", 70), ("uniform", 1), ("uniform", 2500)])
def test_fixed_base_walk_matches_the_plain_walk_and_the_oracle(hostsim, kind, n):
    """the fixed-base form of round 5 (table of 2^(c w) P_i, ONE bucket set for all windows, no closing doublings:
    csrc/bn254_msm.hip) walked with the same lane functions: the same canonical affine point as the plain walk and as the oracle's
    naive multi-exponentiation (gnark-crypto MultiExp under groth16.Prove, gnark-plonky2-verifier/cmd/web-api.go:77)"""
    rng = np.random.default_rng(50 + n)
    pts = cport.bn254_gen_points(n, 5, 3)
    if kind == "witness":
        pts[3] = 0
        pts[n - 1] = 0
    sc = _scalars(kind, n, rng)
    out = (ctypes.c_uint32 * 16)()
    inf = hostsim.hostsim_msm_fixed_g1(pts.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p), n, out)
    got = np.array([out[2 * i] | (out[2 * i + 1] << 32) for i in range(8)], dtype=np.uint64)
    plain, pinf = _run(hostsim, pts, sc)
    assert bool(inf) == pinf and (pinf or np.array_equal(got, plain))
    live = [i for i in range(n) if pts[i].any()]
    want, winf, _ = cport.bn254_msm(pts[live], _reduced(sc[live]), naive=True)
    assert bool(inf) == winf and (winf or np.array_equal(got, want))
