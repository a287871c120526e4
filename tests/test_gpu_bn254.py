"""GPU parity: BN254 G1 MSM through the C ABI vs the oracle (affine result, bit-exact)."""
import numpy as np
import pytest

from oracle import bn254 as bn
from oracle import cport

pytestmark = pytest.mark.gpu


def scalar_words(s):
    return [(s >> (64 * i)) & (2**64 - 1) for i in range(4)]


def rand_scalars(rng, n):
    sc = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)  # < 2^252 < r
    return sc


def unwords(w):
    return (bn.from_mont_words([int(x) for x in w[:4]]), bn.from_mont_words([int(x) for x in w[4:]]))


def test_tiny_against_python(zctx):
    pts = cport.bn254_gen_points(5, 7, 11)
    scalars = [0, 1, bn.R - 1, 123456789, 2**200 + 17]
    sc = np.array([scalar_words(s) for s in scalars], dtype=np.uint64)
    out, inf = zctx.bn254_g1_msm(pts, sc)
    want = bn.msm(scalars, [bn.mul(7 + 11 * i, bn.G1) for i in range(5)])
    assert not inf and unwords(out) == want


def test_scalars_that_are_not_reduced(zctx):
    """a scalar >= r (anything up to 2^256 - 1) is reduced by the kernel instead of losing its top bits (G1 and G2)"""
    scalars = [bn.R, bn.R + 1, 2**256 - 1, 5 * bn.R + 12345, 2**255 + 2**254 + 3, bn.R - 1]
    pts = cport.bn254_gen_points(len(scalars), 7, 11)
    sc = np.array([scalar_words(s) for s in scalars], dtype=np.uint64)
    out, inf = zctx.bn254_g1_msm(pts, sc)
    want = bn.msm([s % bn.R for s in scalars], [bn.mul(7 + 11 * i, bn.G1) for i in range(len(scalars))])
    assert not inf and unwords(out) == want
    out, inf = zctx.bn254_g1_msm(pts[:1], sc[:1])        # r * P = infinity
    assert inf and not out.any()


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1000, 5000, 40000])
def test_uniform_scalars(zctx, n):
    rng = np.random.default_rng(n)
    pts = cport.bn254_gen_points(n, 3, 5) if n else np.zeros((0, 8), np.uint64)
    sc = rand_scalars(rng, n)
    got, ginf = zctx.bn254_g1_msm(pts, sc)
    want, winf, _ = cport.bn254_msm(pts, sc, nthreads=4) if n else (np.zeros(8, np.uint64), True, 1)
    assert ginf == winf and np.array_equal(got, want)


def test_witness_like_distribution(zctx):
    """50% in {0,1}, 30% < 2^64, 20% uniform -> one huge bucket (heavy path)."""
    rng = np.random.default_rng(5)
    n = 30000
    pts = cport.bn254_gen_points(n, 9, 2)
    sc = rand_scalars(rng, n)
    kind = rng.random(n)
    sc[kind < 0.5] = 0
    sc[kind < 0.5, 0] = rng.integers(0, 2, size=int((kind < 0.5).sum()), dtype=np.uint64)
    mid = (kind >= 0.5) & (kind < 0.8)
    sc[mid, 1:] = 0
    got, ginf = zctx.bn254_g1_msm(pts, sc)
    want, winf, _ = cport.bn254_msm(pts, sc, nthreads=4)
    assert ginf == winf and np.array_equal(got, want)


def witness_like(rng, n):
    """50 % in {0, 1}, 30 % < 2^64, 20 % uniform -- what a Groth16 witness vector looks like
    (gnark-plonky2-verifier/cmd/web-api.go:77)"""
    sc = rand_scalars(rng, n)
    kind = rng.random(n)
    small = kind < 0.5
    sc[small] = 0
    sc[small, 0] = rng.integers(0, 2, size=int(small.sum()), dtype=np.uint64)
    mid = (kind >= 0.5) & (kind < 0.8)
    sc[mid, 1:] = 0
    return sc


def test_witness_like_and_adversarial_scalars_at_2_pow_20(zctx):
    """the distributions that matter at (nearly) the headline size, bit-exact against the oracle's C Pippenger: (W) witness-like --
    a quarter of a million scalars equal to 1 land in ONE bucket, cut into ~2 000 slices and summed by the heavy-combine kernel;
    (A1) all scalars equal -- every window has a single bucket holding all 2^20 points; (A2) top windows zero (scalars < 2^64)"""
    n = 1 << 20
    rng = np.random.default_rng(20)
    pts = cport.bn254_gen_points(n, 11, 7)
    cases = {"W": witness_like(rng, n),
             "A1": np.tile(np.array(scalar_words(0x2F0E1D2C3B4A59687766554433221100FFEEDDCCBBAA99887766554433221100 % bn.R), dtype=np.uint64), (n, 1)),
             "A2": rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.array([1, 0, 0, 0], dtype=np.uint64)}
    for name, sc in cases.items():
        sc = np.ascontiguousarray(sc)
        got, ginf = zctx.bn254_g1_msm(pts, sc)
        want, winf, _ = cport.bn254_msm(pts, sc, nthreads=16)
        assert ginf == winf and np.array_equal(got, want), name


def test_adversarial_equal_points_and_scalars(zctx):
    """(A): all points equal (every bucket addition after the first is a DOUBLING), all scalars equal,
    points at infinity in the input, and a cancelling pair."""
    n = 3000
    one = cport.bn254_gen_points(1, 7, 11)
    pts = np.repeat(one, n, axis=0)
    sc = np.tile(np.array(scalar_words(0x1234567890ABCDEF1234567890ABCDEF), dtype=np.uint64), (n, 1))
    got, ginf = zctx.bn254_g1_msm(pts, sc)
    want = bn.mul(n * 0x1234567890ABCDEF1234567890ABCDEF, bn.mul(7, bn.G1))
    assert not ginf and unwords(got) == want
    # infinity inputs are skipped
    pts2 = cport.bn254_gen_points(100, 3, 5)
    pts2[::3] = 0
    sc2 = rand_scalars(np.random.default_rng(1), 100)
    got, ginf = zctx.bn254_g1_msm(pts2, sc2)
    want, winf, _ = cport.bn254_msm(pts2, sc2)
    assert ginf == winf and np.array_equal(got, want)
    # s*P + (r - s)*P = infinity
    two = np.vstack([one, one])
    s2 = np.array([scalar_words(5), scalar_words(bn.R - 5)], dtype=np.uint64)
    got, ginf = zctx.bn254_g1_msm(two, s2)
    assert ginf and not got.any()


def test_linearity_at_2_pow_18(zctx):
    """Full-size property: MSM(a + b) = MSM(a) + MSM(b) on 2^18 points, and agreement with the oracle."""
    n = 1 << 18
    rng = np.random.default_rng(18)
    pts = cport.bn254_gen_points(n, 5, 3)
    a, b = rand_scalars(rng, n), rand_scalars(rng, n)
    a[:, 3] >>= np.uint64(1)
    b[:, 3] >>= np.uint64(1)
    s = a.astype(object) + 0
    ai = [sum(int(a[i, k]) << (64 * k) for k in range(4)) for i in range(0, n, 1)]
    bi = [sum(int(b[i, k]) << (64 * k) for k in range(4)) for i in range(0, n, 1)]
    ab = np.array([scalar_words(x + y) for x, y in zip(ai, bi)], dtype=np.uint64)
    ra, _ = zctx.bn254_g1_msm(pts, a)
    rb, _ = zctx.bn254_g1_msm(pts, b)
    rab, _ = zctx.bn254_g1_msm(pts, ab)
    assert bn.add(unwords(ra), unwords(rb)) == unwords(rab)
    want, _, _ = cport.bn254_msm(pts, a, nthreads=8)
    assert np.array_equal(ra, want)


# ---------------- Poseidon-BN254 hasher (a8)
def _fr_words(x):
    return [(x >> (64 * i)) & (2**64 - 1) for i in range(4)]


def _fr_int(w):
    return sum(int(w[i]) << (64 * i) for i in range(4))


def test_poseidon_bn254_kats_and_random(zctx):
    from oracle import poseidon_bn254 as pb
    import random
    rng = random.Random(2)
    ins = [k["in"] for k in pb.KATS] + [[rng.randrange(pb.R) for _ in range(4)] for _ in range(70)]
    st = np.array([[_fr_words(x) for x in s] for s in ins], dtype=np.uint64)
    out = zctx.poseidon_bn254_permute(st)
    for i, k in enumerate(pb.KATS):   # crypto/plonky2_bn128/src/poseidon_bn128.rs:133-180
        assert [_fr_int(out[i, j]) for j in range(4)] == k["out"]
    for i in range(len(pb.KATS), len(ins)):
        assert [_fr_int(out[i, j]) for j in range(4)] == pb.permute(ins[i])


@pytest.mark.parametrize("width,log_leaves,cap", [(1, 0, 0), (3, 2, 0), (4, 3, 1), (9, 4, 2), (10, 3, 3), (135, 6, 4), (19, 7, 0)])
def test_bn254_merkle_commit(zctx, width, log_leaves, cap):
    from oracle import poseidon_bn254 as pb
    rng = np.random.default_rng(width * 7 + log_leaves)
    GP = 2**64 - 2**32 + 1
    a = rng.integers(0, 2**64, size=(width, 1 << log_leaves), dtype=np.uint64)
    mat = np.where(a >= np.uint64(GP), a - np.uint64(GP), a)
    cap_gpu, levels = zctx.bn254_merkle_commit(mat, cap)
    n = 1 << log_leaves
    layer = [pb.hash_or_noop([int(mat[p, i]) for p in range(width)]) for i in range(n)]
    for l in range(log_leaves - cap + 1):
        assert [_fr_int(d) for d in levels[l]] == layer
        layer = [pb.two_to_one(layer[2 * i], layer[2 * i + 1]) for i in range(len(layer) // 2)]
    assert len(cap_gpu) == 1 << cap


def test_golden_proof_merkle_paths_with_gpu_hashing(zctx):
    """Walk the Merkle paths of the reference's golden proof (near_bft_finality/proofs/random/CGZP...)
    with the GPU kernels only: leaf digests from bn254_merkle_commit, inner nodes from the Poseidon-BN254
    permutation kernel; every path must end in the proof's own cap entry."""
    from conftest import load_golden
    from oracle import plonky2_verifier as V
    j = load_golden("plonky2_near_random_CGZP.json")
    pf = V.parse_proof(j["proof"], j["verifier_data"])
    ch = V.challenges(pf, j["common_data"])
    n_log = j["common_data"]["fri_params"]["degree_bits"] + 3
    caps = [pf["constants_sigmas_cap"], pf["wires_cap"], pf["zs_pp_cap"], pf["quotient_cap"]]

    def gpu_leaf_digest(leaf):
        mat = np.array(leaf, dtype=np.uint64).reshape(-1, 1)      # [width, 1 leaf]
        cap, _ = zctx.bn254_merkle_commit(mat, 0)
        return _fr_int(cap[0])

    def gpu_two_to_one(l, r):
        st = np.array([[_fr_words(0), _fr_words(0), _fr_words(l), _fr_words(r)]], dtype=np.uint64)
        return _fr_int(zctx.poseidon_bn254_permute(st)[0, 0])

    checked = 0
    for rnd, (init, steps) in enumerate(pf["rounds"][:2]):
        x_index = ch["query_indices"][rnd] % (1 << n_log)
        for k in range(4):
            leaf, sib = init[k]
            cur, idx = gpu_leaf_digest(leaf), x_index
            for s in sib:
                cur = gpu_two_to_one(s, cur) if idx & 1 else gpu_two_to_one(cur, s)
                idx >>= 1
            assert cur == caps[k][idx]
            checked += 1
        idx = x_index
        for i, (evals, sib) in enumerate(steps):
            idx >>= 4
            cur, w = gpu_leaf_digest([c for e in evals for c in e]), idx
            for s in sib:
                cur = gpu_two_to_one(s, cur) if w & 1 else gpu_two_to_one(cur, s)
                w >>= 1
            assert cur == pf["commit_caps"][i][w]
            checked += 1
    assert checked == 12


# ---------------------------------------------------------------- Fr NTT (zklc_bn254_fr_ntt)
def _fr_arr(vals):
    from oracle import bn254_fr as FR
    return np.array([FR.to_mont_words(v) for v in vals], dtype=np.uint64)


def _fr_vals(arr):
    from oracle import bn254_fr as FR
    return [FR.from_mont_words(r) for r in arr]


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 6, 10, 12, 13])     # 12, 13: the two-pass LDS path (2^12 .. 2^22, natural order)
def test_fr_ntt_matches_definition(zctx, log_n):
    from oracle import bn254_fr as FR
    import random
    rng = random.Random(log_n)
    n = 1 << log_n
    a = [rng.randrange(FR.R) for _ in range(n)]
    if n <= 64:
        assert FR.ntt(a) == FR.naive_dft(a) and FR.ntt(a, coset=True) == FR.naive_dft(a, FR.GENERATOR)
    got = _fr_vals(zctx.bn254_fr_ntt(_fr_arr(a)))
    assert got == FR.ntt(a)
    assert _fr_vals(zctx.bn254_fr_ntt(_fr_arr(a), coset=1)) == FR.ntt(a, coset=True)
    assert _fr_vals(zctx.bn254_fr_ntt(_fr_arr(a), flags=1)) == FR.ntt(a, inverse=True)
    assert _fr_vals(zctx.bn254_fr_ntt(_fr_arr(a), flags=1, coset=1)) == FR.ntt(a, inverse=True, coset=True)
    if log_n:
        br = [int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)]
        out_br = _fr_vals(zctx.bn254_fr_ntt(_fr_arr(a), flags=4))
        assert [out_br[br[i]] for i in range(n)] == FR.ntt(a)
        in_br = _fr_vals(zctx.bn254_fr_ntt(_fr_arr([a[br[i]] for i in range(n)]), flags=2))
        assert in_br == FR.ntt(a)


def test_fr_ntt_two_pass_equals_stage_path_2p16_2p22(zctx):
    """the two-pass transform against round 1's one-launch-per-stage path (a bit-reversed OUTPUT takes that path: undo the
    reversal) at 2^16 (8 + 8) and 2^22 (11 + 11, the Groth16 size), forward / inverse, with and without the coset"""
    rng = np.random.default_rng(9)
    for log_n in (16, 22):
        n = 1 << log_n
        a = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        idx = np.arange(n, dtype=np.uint64)
        br = np.zeros(n, dtype=np.int64)
        for b in range(log_n):
            br |= (((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(log_n - 1 - b)).astype(np.int64)
        for flags, coset in ((0, 0), (0, 1), (1, 0), (1, 1)):
            fast = zctx.bn254_fr_ntt(a.copy(), flags=flags, coset=coset)
            slow_br = zctx.bn254_fr_ntt(a.copy(), flags=flags | 4, coset=coset)     # OUT_BITREV: the stage path
            assert np.array_equal(fast, slow_br[br]), (log_n, flags, coset)


def test_fr_ntt_roundtrip_and_coset_property_2p18(zctx):
    """size-independent properties at a large size: inverse(forward(x)) == x; coset evaluation of the vanishing polynomial
    X^n - 1 is the constant 5^n - 1 (what the Groth16 quotient step divides by)"""
    from oracle import bn254_fr as FR
    log_n = 18
    n = 1 << log_n
    rng = np.random.default_rng(5)
    a = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)                             # < 2^252 < r: canonical Montgomery images
    fwd = zctx.bn254_fr_ntt(a, coset=1)
    back = zctx.bn254_fr_ntt(fwd, flags=1, coset=1)
    assert np.array_equal(back, a)
    x = np.zeros((n, 4), dtype=np.uint64)
    x[0] = FR.to_mont_words(FR.R - 1)          # -1 + X^n wraps onto the constant term in a size-n transform: use 2n
    big = np.zeros((2 * n, 4), dtype=np.uint64)
    big[0] = FR.to_mont_words(FR.R - 1)
    big[n] = FR.to_mont_words(1)
    ev = _fr_vals(zctx.bn254_fr_ntt(big, coset=1)[:8])
    g2n = pow(FR.GENERATOR, n, FR.R)
    w = FR.root(log_n + 1)
    assert ev == [(g2n * pow(w, k * n, FR.R) - 1) % FR.R for k in range(8)]


# ---------------------------------------------------------------- G2 MSM (zklc_bn254_g2_msm)
def _g2_points(n, seed):
    import random
    from oracle import bn254 as B
    rng = random.Random(seed)
    pts, cur = [], B.g2_mul(rng.randrange(1, B.R), B.G2)
    step = B.g2_mul(rng.randrange(1, B.R), B.G2)
    for _ in range(n):
        pts.append(cur)
        cur = B.g2_add(cur, step)
    return pts


def test_g2_msm_small_against_python(zctx):
    import random
    from oracle import bn254 as B
    rng = random.Random(21)
    for n in [0, 1, 2, 3, 17, 70]:
        pts = _g2_points(n, n)
        sc = [rng.randrange(B.R) for _ in range(n)]
        if n >= 3:
            sc[0], sc[1] = 0, 1
            pts[2] = None                                  # a point at infinity in the input
        want = B.g2_msm(sc, pts)
        pa = np.array([B.g2_to_words(p) for p in pts], dtype=np.uint64).reshape(-1, 16)
        sa = np.array([[(s >> (64 * i)) & (2**64 - 1) for i in range(4)] for s in sc], dtype=np.uint64).reshape(-1, 4)
        out, inf = zctx.bn254_g2_msm(pa, sa)
        assert B.g2_from_words([int(x) for x in out], inf) == want, n


def test_g2_msm_adversarial_and_linearity(zctx):
    """all points equal / P and -P cancelling (exceptional cases of the bucket additions), and at 2^12 points the
    size-independent identity MSM(s, Q) + MSM(t, Q) == MSM(s + t, Q)"""
    import random
    from oracle import bn254 as B
    rng = random.Random(22)
    q = B.g2_mul(777, B.G2)
    n = 300
    sc = [rng.randrange(B.R) for _ in range(n)]
    pa = np.array([B.g2_to_words(q)] * n, dtype=np.uint64)
    sa = np.array([[(s >> (64 * i)) & (2**64 - 1) for i in range(4)] for s in sc], dtype=np.uint64)
    out, inf = zctx.bn254_g2_msm(pa, sa)
    assert B.g2_from_words([int(x) for x in out], inf) == B.g2_mul(sum(sc) % B.R, q)
    pa2 = np.array([B.g2_to_words(q), B.g2_to_words(B.g2_neg(q))] * 8, dtype=np.uint64)
    sa2 = np.array([[5, 0, 0, 0]] * 16, dtype=np.uint64)
    out, inf = zctx.bn254_g2_msm(pa2, sa2)
    assert inf and not out.any()
    n = 1 << 12
    pts = _g2_points(n, 99)
    pa = np.array([B.g2_to_words(p) for p in pts], dtype=np.uint64)
    s = [rng.randrange(B.R) for _ in range(n)]
    t = [rng.randrange(B.R) for _ in range(n)]
    words = lambda v: np.array([[(x >> (64 * i)) & (2**64 - 1) for i in range(4)] for x in v], dtype=np.uint64)
    a = B.g2_from_words([int(x) for x in zctx.bn254_g2_msm(pa, words(s))[0]])
    b = B.g2_from_words([int(x) for x in zctx.bn254_g2_msm(pa, words(t))[0]])
    c = B.g2_from_words([int(x) for x in zctx.bn254_g2_msm(pa, words([(x + y) % B.R for x, y in zip(s, t)]))[0]])
    assert B.g2_add(a, b) == c and B.g2_is_on_curve(c)


# ---------------------------------------------------------------- pairing (zklc_bn254_pairing_check)
def _g1_words(p):
    from oracle import bn254 as B
    return [0] * 8 if p is None else B.to_mont_words(p[0]) + B.to_mont_words(p[1])


def _gt_from_words(w):
    from oracle import bn254 as B
    v = [B.from_mont_words([int(x) for x in w[4 * i:4 * i + 4]]) for i in range(12)]
    c = [(v[2 * i], v[2 * i + 1]) for i in range(6)]
    return ((c[0], c[1], c[2]), (c[3], c[4], c[5]))


def test_pairing_value_bilinearity_and_infinity(zctx):
    from oracle import bn254 as B
    from oracle import bn254_pairing as PR
    p, q = B.mul(0xABCDEF, B.G1), B.g2_mul(0x13579B, B.G2)
    checks = [[(p, q)], [(p, q), (B.neg(p), q)], [(None, q), (p, None)],
              [(B.mul(5, B.G1), B.G2), (B.neg(B.G1), B.g2_mul(5, B.G2))], [(B.G1, B.G2), (B.G1, B.G2)]]
    k = 2
    g1 = np.array([[_g1_words(c[i][0]) if i < len(c) else [0] * 8 for i in range(k)] for c in checks], dtype=np.uint64)
    g2 = np.array([[B.g2_to_words(c[i][1]) if i < len(c) else [0] * 16 for i in range(k)] for c in checks], dtype=np.uint64)
    ok, gt = zctx.bn254_pairing_check(g1, g2, k, want_gt=True)
    assert ok.tolist() == [0, 1, 1, 1, 0]
    assert _gt_from_words(gt[0]) == PR.pairing(p, q)
    assert _gt_from_words(gt[1]) == PR.F12_ONE
    e = PR.pairing(B.G1, B.G2)
    assert _gt_from_words(gt[4]) == PR.f12_mul(e, e)


def test_reference_groth16_kat_on_the_gpu(zctx):
    """the reference's Groth16 proof (contracts/hardhat/test/proof_with_witness.json) verifies against the verifying key of
    Verifier.sol:57-83 through the GPU pairing; the tampered vectors of test/verify.ts do not; batch of 256 checks"""
    from conftest import load_golden
    from oracle import bn254 as B
    from test_oracle_pairing import kat_vk
    j = load_golden("groth16_kat.json")
    vk = kat_vk(j)

    def pairs(proof, inputs):
        a = (proof[0], proof[1])
        b = ((proof[3], proof[2]), (proof[5], proof[4]))
        c = (proof[6], proof[7])
        l = vk["ic"][0]
        for s, pt in zip(inputs, vk["ic"][1:]):
            l = B.add(l, B.mul(s, pt))
        return [(a, b), (c, vk["delta_neg"]), (vk["alpha"], vk["beta_neg"]), (l, vk["gamma_neg"])]
    proof, inputs = [int(x) for x in j["proof"]], [int(x) for x in j["inputs"]]
    good = pairs(proof, inputs)
    bad_in = pairs(proof, [int(x) for x in j["incorrect_inputs"]])
    checks = [good, bad_in] * 128
    g1 = np.array([[_g1_words(p) for p, _ in c] for c in checks], dtype=np.uint64)
    g2 = np.array([[B.g2_to_words(q) for _, q in c] for c in checks], dtype=np.uint64)
    ok, _ = zctx.bn254_pairing_check(g1, g2, 4)
    assert ok.tolist() == [1, 0] * 128


# ---------------------------------------------------------------- fixed-base form (zklc_bn254_g{1,2}_msm_fixed_*)
def _fixed_msm(zctx, pts, sc, group=1):
    """table built once on the device, then the multi-exponentiation over it: -> (affine words, is_infinity)"""
    import torch
    n, aff = pts.shape[0], 8 * group
    dev = "cuda:%d" % zctx.device_id
    d_pts = torch.from_numpy(np.ascontiguousarray(pts).view(np.int64)).to(dev)
    d_sc = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).to(dev)
    table = zctx.bn254_msm_fixed_table(d_pts, n, group=group)
    wb = zctx.bn254_g1_msm_workspace_bytes(n) if group == 1 else int(zctx._lib.zklc_bn254_g2_msm_workspace_bytes(n))
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(aff, dtype=torch.int64, device=dev)
    d_inf = torch.zeros(1, dtype=torch.int32, device=dev)
    outs = []
    for _ in range(2):                      # the table is reusable: the second call must give the same point
        zctx.bn254_msm_fixed_dev(table, d_sc, n, d_out, d_inf, ws, wb, group=group)
        torch.cuda.synchronize()
        outs.append((d_out.cpu().numpy().view(np.uint64).copy(), bool(int(d_inf[0]))))
    assert outs[0][1] == outs[1][1] and np.array_equal(outs[0][0], outs[1][0])
    return outs[0]


@pytest.mark.parametrize("n", [1, 2, 65, 1000, 5000, 40000, 1 << 17])
def test_fixed_base_g1_equals_the_plain_form_and_the_oracle(zctx, n):
    """the fixed-base form (table of 2^(c w) P_i, one bucket set for all windows) gives the canonical affine point of the plain form
    bit for bit: uniform scalars, points at infinity among the bases, zero scalars and scalars >= r (web-api.go:77: the proving
    key's bases are the same for every proof)"""
    rng = np.random.default_rng(n + 7)
    pts = cport.bn254_gen_points(n, 3, 5)
    sc = rand_scalars(rng, n)
    if n >= 65:
        pts[::17] = 0                                           # infinity among the bases
        sc[5] = 0
        sc[6] = np.array(scalar_words(bn.R + 12345), dtype=np.uint64)
        sc[7] = np.array(scalar_words(2**256 - 1), dtype=np.uint64)
    got, ginf = _fixed_msm(zctx, pts, sc)
    plain, pinf = zctx.bn254_g1_msm(pts, sc)
    assert ginf == pinf and np.array_equal(got, plain)
    red = sc.copy()
    if n >= 65:
        red[6] = np.array(scalar_words(12345), dtype=np.uint64)
        red[7] = np.array(scalar_words((2**256 - 1) % bn.R), dtype=np.uint64)
    want, winf, _ = cport.bn254_msm(pts, red, nthreads=8)
    assert ginf == winf and np.array_equal(got, want)


def test_fixed_base_g1_distributions_at_2_pow_20(zctx):
    """witness-like (one bucket of a quarter of a million entries per window -> with ONE bucket set, four million: the heavy-combine
    path), all scalars equal, scalars below 2^64: fixed-base == the oracle's Pippenger at 2^20"""
    n = 1 << 20
    rng = np.random.default_rng(21)
    pts = cport.bn254_gen_points(n, 11, 7)
    cases = {"W": witness_like(rng, n),
             "A1": np.tile(np.array(scalar_words(0x2F0E1D2C3B4A59687766554433221100FFEEDDCCBBAA99887766554433221100 % bn.R), dtype=np.uint64), (n, 1)),
             "A2": rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.array([1, 0, 0, 0], dtype=np.uint64)}
    for name, sc in cases.items():
        sc = np.ascontiguousarray(sc)
        got, ginf = _fixed_msm(zctx, pts, sc)
        want, winf, _ = cport.bn254_msm(pts, sc, nthreads=16)
        assert ginf == winf and np.array_equal(got, want), name


def test_fixed_base_table_is_checked(zctx):
    """a table of another size (or no table at all) is refused with ZKLC_ERR_INVALID_ARG, not read"""
    import torch
    import zklc_amd
    n = 300
    dev = "cuda:%d" % zctx.device_id
    pts = cport.bn254_gen_points(n, 3, 5)
    d_pts = torch.from_numpy(pts.view(np.int64)).to(dev)
    table = zctx.bn254_msm_fixed_table(d_pts, n)
    d_sc = torch.zeros((2 * n, 4), dtype=torch.int64, device=dev)
    wb = zctx.bn254_g1_msm_workspace_bytes(2 * n)
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    d_out, d_inf = torch.zeros(8, dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    with pytest.raises(zklc_amd.ZklcError):
        zctx.bn254_msm_fixed_dev(table, d_sc, 2 * n, d_out, d_inf, ws, wb)          # n does not match the table
    with pytest.raises(zklc_amd.ZklcError):
        zctx.bn254_msm_fixed_dev(table, d_sc, n, d_out, d_inf, ws, wb, group=2)     # a G1 table handed to the G2 entry point
    junk = torch.zeros(4096, dtype=torch.uint8, device=dev)
    junk = junk[(-junk.data_ptr()) % 256:]
    with pytest.raises(zklc_amd.ZklcError):
        zctx.bn254_msm_fixed_dev(junk, d_sc, n, d_out, d_inf, ws, wb)


def test_fixed_base_g2_equals_the_plain_form_and_the_oracle(zctx):
    """G2 fixed-base multi-exponentiation == the plain form (same canonical affine words) == the ORACLE (oracle/bn254.py g2_msm, pinned by
    the Groth16 KAT), with zero scalars and a base at infinity"""
    import random
    from oracle import bn254 as B
    for n in (1, 9, 300):
        pts = _g2_points(n, 100 + n)
        rng = random.Random(n)
        sc = [rng.randrange(B.R) for _ in range(n)]
        if n > 4:
            sc[2] = 0
            pts[3] = None                                       # infinity among the bases
        pa = np.array([B.g2_to_words(p) for p in pts], dtype=np.uint64).reshape(-1, 16)
        sa = np.array([scalar_words(s) for s in sc], dtype=np.uint64)
        got, ginf = _fixed_msm(zctx, pa, sa, group=2)
        plain, pinf = zctx.bn254_g2_msm(pa, sa)
        assert ginf == pinf and np.array_equal(got, plain), n
        assert B.g2_from_words([int(x) for x in got], ginf) == B.g2_msm(sc, pts), n
