// TEST INFRASTRUCTURE: g++ build of the *.cuh arithmetic headers so the CPU
// test-suite (pytest -m "not gpu") can check the exact kernel arithmetic
// against the oracle without a GPU.  Never linked into libzklc_mi355.so.
#include "../../zk-light-client-implementation_amd/csrc/ed25519_verify.cuh"
#include "../../zk-light-client-implementation_amd/csrc/poseidon_gl.cuh"
#include "../../zk-light-client-implementation_amd/csrc/goldilocks_ntt_group.cuh"
#include "../../zk-light-client-implementation_amd/csrc/bn254_msm_lane.cuh"
#include "../../zk-light-client-implementation_amd/csrc/poseidon_bn254.cuh"
#include "../../zk-light-client-implementation_amd/csrc/plonky2_gates.cuh"
#include "../../zk-light-client-implementation_amd/csrc/bn254_pairing.cuh"
#include "../../zk-light-client-implementation_amd/csrc/bn254_fr_ntt_tile.cuh"
#include <string.h>
#include <vector>

typedef ec_xyzz<FpField> g1_xyzz;
typedef ec_xyzz<Fp2Field> g2_xyzz;

static ge_niels g_btab[ZKLC_ED_BTABLE];
static int g_btab_ready = 0;

extern "C" {

void hostsim_fe_op(int op, const u32 *a, const u32 *b, u32 *out) {
    fe x = fe_from_words(a), y = fe_from_words(b), r;  // inputs < 2^255
    switch (op) {
        case 0: r = fe_add(x, y); break;
        case 1: r = fe_sub(x, y); break;
        case 2: r = fe_mul(x, y); break;
        case 3: r = fe_sqr(x); break;
        case 4: r = fe_invert(x); break;
        case 5: r = fe_pow22523(x); break;
        case 6: r = fe_freeze(x); break;
        case 7: r = fe_sqr2(x); break;
        case 8: {  // three-term lazy inputs built from REDUCED values
            fe xr = fe_mul(x, fe_one()), yr = fe_mul(y, fe_one());
            r = fe_mul(fe_add(fe_add(xr, yr), xr), fe_sub(fe_sub(yr, xr), xr));
            break;
        }
        default: r = fe_zero();
    }
    fe_freeze_words(out, r);  // canonical
}

void hostsim_sc_reduce512(const u32 *x, u32 *out) { sc_reduce512(out, x); }
u32 hostsim_sc_is_canonical(const u32 *x) { return sc_is_canonical(x); }

void hostsim_sha512(const uint8_t *msg, u32 len, uint8_t *out) {
    u64 h[8];
    sha512_hash_msg(msg, len, h);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h[i] >> (56 - 8 * j));
}

u32 hostsim_decompress_compress(const uint8_t *in, uint8_t *out) {
    u32 w[8];
    memcpy(w, in, 32);
    ge_p3 p;
    u32 ok = ge_decompress(p, w);
    u32 o[8];
    ge_compress(o, p);
    memcpy(out, o, 32);
    return ok;
}

void hostsim_base_table(u32 *out /*128*30 words*/) {
    if (!g_btab_ready) {
        for (u32 j = 1; j <= ZKLC_ED_BTABLE; j++) g_btab[j - 1] = ed25519_base_table_entry(j);
        g_btab_ready = 1;
    }
    memcpy(out, g_btab, sizeof(g_btab));
}

u32 hostsim_ed25519_verify(const uint8_t *pk, const uint8_t *sig, const uint8_t *msg, u32 msg_len) {
    if (!g_btab_ready) {
        u32 tmp[ZKLC_ED_BTABLE * 30];
        hostsim_base_table(tmp);
    }
    u32 pkw[8], sigw[16];
    memcpy(pkw, pk, 32);
    memcpy(sigw, sig, 64);
    i32 tab[ZKLC_ED_ATAB_WORDS];
    return ed25519_verify_one<1>(pkw, sigw, msg, msg_len, g_btab, tab);
}

u64 hostsim_gl_op(int op, u64 a, u64 b) {
    switch (op) {
        case 0: return gl_add(a, b);
        case 1: return gl_sub(a, b);
        case 2: return gl_mul(a, b);
        case 3: return gl_inv(a);
        case 4: return gl_reduce128(a, b);
        case 5: return gl_root_of_unity((u32)a);
        case 6: return gl_reduce128_loose(a, b);
        case 7: return gl_add_lc(a, b);
        case 8: return gl_mul_loose(a, b);
        case 9: return gl_canonical(a);
        default: return 0;
    }
}
// sum_i x[i] * y[i] through the 160-bit accumulator
u64 hostsim_gl_acc(const u64 *x, const u64 *y, u32 n) {
    gl_acc160 acc = {0, 0, 0};
    for (u32 i = 0; i < n; i++) gl_acc_mul(acc, x[i], y[i]);
    return gl_acc_reduce(acc);
}
// sum_i x[i] * k[i] through the carry-free three-column accumulator and the 22-bit limb table (gl_acc3 / gl_limbs22): the
// accumulation of the quotient kernels' p2_consumer and of p2_fri_combine_kernel; `fold_every` terms between normalisations
u64 hostsim_gl_acc3(const u64 *x, const u64 *k, u32 n, u32 fold_every) {
    gl_acc3 acc = {0, 0, 0};
    u32 t6[6];
    for (u32 i = 0; i < n; i++) {
        if (fold_every && i && i % fold_every == 0) gl_acc3_normalize(acc);
        gl_limbs22(k[i], t6);
        gl_acc3_mul(acc, x[i], (const u32 *)t6);
    }
    return gl_acc3_reduce(acc);
}
void hostsim_poseidon_gl_permute(u64 *s) { poseidon_gl_permute(s); }
void hostsim_poseidon_gl_hash(const u64 *in, u32 len, u64 *out4) { poseidon_gl_hash_or_noop(in, 1, len, out4); }
void hostsim_poseidon_gl_two_to_one(const u64 *l, const u64 *r, u64 *out4) { poseidon_gl_two_to_one(l, r, out4); }

// BN254: operands and results in gnark Montgomery words (8 x u32 per Fp element)
void hostsim_fp_op(int op, const u32 *a, const u32 *b, u32 *out) {
    fp xl = fp_from_gnark(a), yl = fp_from_gnark(b), x = fp_reduce(xl), y = fp_reduce(yl), r;
    switch (op) {
        case 0: r = fp_add(x, y); break;
        case 1: r = fp_sub(x, y); break;
        case 2: r = fp_mul(xl, yl); break;  // lazy operands straight into the multiplier
        case 3: r = fp_sqr(xl); break;
        case 4: r = fp_inv(x); break;
        case 5: r = fp_mul(fp_sub(fp_sub(fp_mul(x, y), x), fp_dbl(y)), fp_add(fp_add(x, y), fp_mul(x, x))); break;  // lazy chains
        default: r = fp_zero();
    }
    fp_to_gnark(out, r);
}
// op 0: P + Q (mixed), 1: P - Q (mixed), 2: 2P, 3: (P + Q) + (P + Q) general add, 4: ((P+Q)+Q)+... chain of n mixed adds of Q
u32 hostsim_g1_op(int op, const u32 *p16, u32 pinf, const u32 *q16, u32 qinf, u32 n, u32 *out16) {
    fp qx = fp_from_gnark(q16), qy = fp_from_gnark(q16 + 8);      // Q stays lazy, as in the MSM bucket loop
    g1_xyzz a, r;
    a.X = fp_reduce(fp_from_gnark(p16));
    a.Y = fp_reduce(fp_from_gnark(p16 + 8));
    a.ZZ = pinf ? fp_zero() : FpField::one();
    a.ZZZ = a.ZZ;
    switch (op) {
        case 0: r = qinf ? a : ec_add_affine<FpField>(a, qx, qy, 0); break;
        case 1: r = qinf ? a : ec_add_affine<FpField>(a, qx, qy, 1); break;
        case 2: r = ec_double(a); break;
        case 3: { g1_xyzz s = qinf ? a : ec_add_affine<FpField>(a, qx, qy, 0); r = ec_add(s, s); break; }
        case 4: { r = a; for (u32 i = 0; i < n; i++) r = ec_add_affine<FpField>(r, qx, qy, 0); break; }
        case 5: {
            g1_xyzz s;
            s.X = fp_reduce(qx);
            s.Y = fp_reduce(qy);
            s.ZZ = qinf ? fp_zero() : FpField::one();
            s.ZZZ = s.ZZ;
            r = ec_add(a, s);
            break;
        }
        default: r = ec_infinity<FpField>();
    }
    return ec_to_affine_gnark(out16, r);
}

// The whole multi-scalar multiplication of csrc/bn254_msm.hip as a sequential walk over its lanes: recode -> (chunk, window) tile
// histograms -> prefixes / totals / offsets -> scatter -> mode 0 (the product): converted points, one "lane" per SLICE of the sorted
// entries (msm_slice_lane), then the combine of the buckets cut by slice boundaries (buckets cut into more than `heavy_min` slices
// take the strided form of the heavy-combine kernel with `heavy_threads` partial sums); mode 1: one lane per bucket on the raw
// points (msm_bucket_lane) -> segments -> windows -> doublings.  Same functions,
// same plan, same digit codes as the kernels; only the parallel glue (LDS atomics, tree reductions) is replaced by loops.
}  // extern "C"
template <class F>
static u32 hostsim_msm(const u64 *points, const u64 *scalars, u32 n, u32 mode, u32 heavy_min, u32 heavy_threads, u32 *out) {
    const int AFF = msm_cfg<F>::AFF, XY = msm_cfg<F>::XYZZ;
    // mode 0: the product plan (G1: the endomorphism split from MSM_GLV_MIN_POINTS points on); modes 1 / 2: no split
    msm_plan pl = msm_make_plan(n, F::LIMBS == 10 && mode == 0);
    const u32 bpw = pl.buckets_per_window;
    std::vector<unsigned short> dig((size_t)pl.n_pad * pl.windows, 0);
    std::vector<i32> glv_points;                      // msm_glv_prepare_kernel: digits of both halves + the two point records
    if (pl.glv) {
        const int W = msm_rec<F, true>::WORDS;
        glv_points.resize((size_t)pl.n * W);
        for (u32 i = 0; i < n; i++) {
            bool live = !msm_point_is_inf<AFF>(points, i);
            u32 sw[8], m1[8] = {0}, m2[8] = {0}, neg1 = 0, neg2 = 0;
            if (live) {
                msm_load_scalar(scalars, i, sw);
                msm_glv_split(sw, m1, neg1, m2, neg2);
            }
            u32 carry1 = 0, carry2 = 0;
            for (u32 w = 0; w < pl.windows; w++) {
                int d1 = live ? msm_digit(m1, w, pl.c, carry1) : 0, d2 = live ? msm_digit(m2, w, pl.c, carry2) : 0;
                dig[(size_t)w * pl.n_pad + i] = (unsigned short)msm_digit_code(d1);
                dig[(size_t)w * pl.n_pad + n + i] = (unsigned short)msm_digit_code(d2);
            }
            if (carry1 | carry2) return 0xdeadu;      // the top window never carries out (|k_i| < 2^127)
            msm_convert_point_glv<F, true>(glv_points.data() + (size_t)i * W, glv_points.data() + ((size_t)n + i) * W, points, i, neg1, neg2);
        }
    } else {
        for (u32 i = 0; i < pl.n_pad; i++) {
            bool live = i < pl.n && !msm_point_is_inf<AFF>(points, i);
            u32 sw[8], carry = 0;
            if (live) msm_load_scalar(scalars, i, sw);
            for (u32 w = 0; w < pl.windows; w++) {
                int d = live ? msm_digit(sw, w, pl.c, carry) : 0;
                dig[(size_t)w * pl.n_pad + i] = (unsigned short)msm_digit_code(d);
            }
        }
    }
    std::vector<u32> cnt((size_t)pl.total_buckets * pl.chunks, 0), totals(pl.total_buckets), offsets(pl.total_buckets);
    for (u32 w = 0; w < pl.windows; w++)
        for (u32 k = 0; k < pl.chunks; k++) {
            u32 lo = k * pl.chunk_len, hi = lo + pl.chunk_len < pl.n_pad ? lo + pl.chunk_len : pl.n_pad;
            for (u32 i = lo; i < hi; i++) {
                u32 code = dig[(size_t)w * pl.n_pad + i], neg;
                if (code) cnt[((size_t)w * pl.chunks + k) * bpw + msm_code_bucket(code, neg)]++;
            }
        }
    for (u32 key = 0; key < pl.total_buckets; key++) {
        u32 w = key / bpw, b = key % bpw, run = 0;
        for (u32 k = 0; k < pl.chunks; k++) {
            u32 &c = cnt[((size_t)w * pl.chunks + k) * bpw + b];
            u32 t = c;
            c = run;
            run += t;
        }
        totals[key] = run;
    }
    u32 run = 0;
    for (u32 key = 0; key < pl.total_buckets; key++) {
        offsets[key] = run;
        run += totals[key];
    }
    std::vector<u32> entries((size_t)run + 1, 0);
    for (u32 w = 0; w < pl.windows; w++)
        for (u32 k = 0; k < pl.chunks; k++) {
            std::vector<u32> cur(bpw);
            for (u32 b = 0; b < bpw; b++) cur[b] = offsets[(size_t)w * bpw + b] + cnt[((size_t)w * pl.chunks + k) * bpw + b];
            u32 lo = k * pl.chunk_len, hi = lo + pl.chunk_len < pl.n_pad ? lo + pl.chunk_len : pl.n_pad;
            for (u32 i = lo; i < hi; i++) {
                u32 code = dig[(size_t)w * pl.n_pad + i], neg;
                if (!code) continue;
                u32 b = msm_code_bucket(code, neg);
                entries[cur[b]++] = (i << 1) | neg;
            }
        }
    std::vector<i32> buckets((size_t)pl.total_buckets * XY, 0);       // zeroed = infinity, as the product memsets it
    if (mode != 1) {
        // the product path: converted points, slices of the sorted entries, combine
        const u32 E = run, slices = (u32)(((u64)pl.n * pl.windows + MSM_SLICE - 1) / MSM_SLICE);
        std::vector<i32> cpoints((size_t)pl.n * 2 * F::LIMBS), partials(((size_t)slices + 1) * 2 * XY, 0x5a5a5a5a);
        if (mode == 0) {        // packed 8 x 32-bit records (the product default)
            if (pl.glv)
                cpoints = glv_points;
            else
                for (u32 i = 0; i < n; i++) msm_convert_point<F, true>(cpoints.data() + (size_t)i * msm_rec<F, true>::WORDS, points, i);
            for (u32 lane = 0; lane < slices; lane++)
                msm_slice_lane<F, true>(cpoints.data(), entries.data(), offsets.data(), pl.total_buckets, E, lane, buckets.data(), partials.data());
        } else {                // mode 2: records of ten 32-bit limbs per coordinate (ZKLC_MSM_PACKED=0)
            for (u32 i = 0; i < n; i++) msm_convert_point<F, false>(cpoints.data() + (size_t)i * 2 * F::LIMBS, points, i);
            for (u32 lane = 0; lane < slices; lane++)
                msm_slice_lane<F, false>(cpoints.data(), entries.data(), offsets.data(), pl.total_buckets, E, lane, buckets.data(), partials.data());
        }
        for (u32 key = 0; key < pl.total_buckets; key++) {
            u32 la, lb, which;
            if (!msm_combine_span<F>(offsets.data(), totals.data(), key, la, lb, which)) continue;
            if (lb - la + 1 > heavy_min) {      // the heavy-combine kernel: strided partial sums, then their sum
                ec_xyzz<F> acc = ec_infinity<F>();
                for (u32 t = 0; t < heavy_threads; t++) {
                    ec_xyzz<F> part = ec_infinity<F>();
                    for (u32 j = la + t; j <= lb; j += heavy_threads)
                        part = ec_add(part, msm_load_xyzz<F>(partials.data() + ((size_t)2 * j + (j == la ? which : 0u)) * XY));
                    acc = ec_add(acc, part);
                }
                msm_store_xyzz<F>(buckets.data() + (size_t)key * XY, acc);
            } else {
                msm_combine_lane<F>(offsets.data(), totals.data(), key, partials.data(), buckets.data());
            }
        }
    } else {
        // one lane per bucket on the raw points (msm_bucket_lane; the strided form above `heavy_min` entries)
        for (u32 key = 0; key < pl.total_buckets; key++) {
            ec_xyzz<F> acc = ec_infinity<F>();
            if (totals[key] >= heavy_min) {
                for (u32 t = 0; t < heavy_threads; t++) {
                    ec_xyzz<F> part = ec_infinity<F>();
                    msm_bucket_lane<F>(part, points, entries.data(), offsets[key], t, totals[key], heavy_threads);
                    acc = ec_add(acc, part);
                }
            } else {
                msm_bucket_lane<F>(acc, points, entries.data(), offsets[key], 0, totals[key], 1);
            }
            msm_store_xyzz<F>(buckets.data() + (size_t)key * XY, acc);
        }
    }
    u32 seg_per_window = (bpw + MSM_SEG - 1) / MSM_SEG;
    ec_xyzz<F> total = ec_infinity<F>();
    for (u32 w = pl.windows; w-- > 0;) {
        ec_xyzz<F> win = ec_infinity<F>();
        for (u32 si = 0; si < seg_per_window; si++) win = ec_add(win, msm_segment_lane<F>(buckets.data(), pl, w, si));
        if (mode != 1) {                                              // as msm_final_kernel (G1 and G2): the doublings by a quad of lanes
            ec_xyzz<F> q4[4] = {win, win, win, win};
            for (u32 k = 0; k < pl.c * w; k++) ec_double_quad_ref<F>(q4);
            win = q4[0];
        } else {
            for (u32 k = 0; k < pl.c * w; k++) win = ec_double(win);  // 2^(c w) * window_w, then the sum
        }
        total = ec_add(total, win);
    }
    return ec_to_affine_gnark(out, total);
}
// The FIXED-BASE form of the multi-exponentiation (csrc/bn254_msm.hip, round 5) walked on the CPU with the same lane functions: the
// table of 2^(c w) P_i as packed affine records (what msm_fixed_table_kernel writes: c doublings + one inversion per row), the digits
// of all windows sorted into ONE bucket set (the (chunk, window) tiles of the sort become the tiles of a single window; an entry
// names table row w * n + i), slices, combine, ONE bucket reduction and no closing doublings.
template <class F>
static u32 hostsim_msm_fixed(const u64 *points, const u64 *scalars, u32 n, u32 *out) {
    typedef typename F::T T;
    const int AFF = msm_cfg<F>::AFF, XY = msm_cfg<F>::XYZZ, W = msm_rec<F, true>::WORDS;
    msm_plan pl = msm_make_plan(n, false);
    const u32 bpw = pl.buckets_per_window;
    std::vector<i32> table((size_t)pl.windows * n * W, 0);
    for (u32 i = 0; i < n; i++) {
        if (msm_point_is_inf<AFF>(points, i)) continue;              // all-zero records
        T x, y;
        msm_load_point<F>(points, i, x, y);
        x = F::reduce(x);
        y = F::reduce(y);
        for (u32 w = 0; w < pl.windows; w++) {
            u32 *dst = reinterpret_cast<u32 *>(table.data() + ((size_t)w * n + i) * W);
            F::pack(dst, x);
            F::pack(dst + F::PACKW, y);
            if (w + 1 == pl.windows) break;
            ec_xyzz<F> acc;
            acc.X = x;
            acc.Y = y;
            acc.ZZ = acc.ZZZ = F::one();
            for (u32 k = 0; k < pl.c; k++) acc = ec_double(acc);
            T inv = F::inv(F::mul(acc.ZZ, acc.ZZZ));
            x = F::reduce(F::mul(acc.X, F::mul(inv, acc.ZZZ)));
            y = F::reduce(F::mul(acc.Y, F::mul(inv, acc.ZZ)));
        }
    }
    std::vector<unsigned short> dig((size_t)pl.windows * pl.n_pad, 0);
    for (u32 i = 0; i < pl.n_pad; i++) {
        bool live = i < pl.n;
        if (live) {
            i32 acc = 0;
            for (int k = 0; k < W; k++) acc |= table[(size_t)i * W + k];
            live = acc != 0;
        }
        u32 sw[8], carry = 0;
        if (live) msm_load_scalar(scalars, i, sw);
        for (u32 w = 0; w < pl.windows; w++) {
            int d = live ? msm_digit(sw, w, pl.c, carry) : 0;
            dig[(size_t)w * pl.n_pad + i] = (unsigned short)msm_digit_code(d);
        }
    }
    msm_plan pm = pl;                                                // the view behind the sort: one window, windows x chunks tiles
    pm.windows = 1;
    pm.chunks = pl.windows * pl.chunks;
    pm.total_buckets = bpw;
    std::vector<u32> cnt((size_t)bpw * pm.chunks, 0), totals(bpw), offsets(bpw);
    for (u32 w = 0; w < pl.windows; w++)
        for (u32 k = 0; k < pl.chunks; k++) {
            u32 lo = k * pl.chunk_len, hi = lo + pl.chunk_len < pl.n_pad ? lo + pl.chunk_len : pl.n_pad;
            for (u32 i = lo; i < hi; i++) {
                u32 code = dig[(size_t)w * pl.n_pad + i], neg;
                if (code) cnt[((size_t)w * pl.chunks + k) * bpw + msm_code_bucket(code, neg)]++;
            }
        }
    for (u32 b = 0; b < bpw; b++) {                                  // msm_totals_kernel with the merged plan
        u32 run = 0;
        for (u32 t = 0; t < pm.chunks; t++) {
            u32 &c = cnt[(size_t)t * bpw + b];
            u32 v = c;
            c = run;
            run += v;
        }
        totals[b] = run;
    }
    u32 run = 0;
    for (u32 b = 0; b < bpw; b++) {
        offsets[b] = run;
        run += totals[b];
    }
    std::vector<u32> entries((size_t)run + 1, 0);
    for (u32 w = 0; w < pl.windows; w++)
        for (u32 k = 0; k < pl.chunks; k++) {
            std::vector<u32> cur(bpw);
            for (u32 b = 0; b < bpw; b++) cur[b] = offsets[b] + cnt[((size_t)w * pl.chunks + k) * bpw + b];
            u32 lo = k * pl.chunk_len, hi = lo + pl.chunk_len < pl.n_pad ? lo + pl.chunk_len : pl.n_pad;
            for (u32 i = lo; i < hi; i++) {
                u32 code = dig[(size_t)w * pl.n_pad + i], neg;
                if (!code) continue;
                u32 b = msm_code_bucket(code, neg);
                entries[cur[b]++] = ((w * n + i) << 1) | neg;
            }
        }
    std::vector<i32> buckets((size_t)bpw * XY, 0);
    const u32 E = run, slices = (u32)(((u64)pl.n * pl.windows + MSM_SLICE - 1) / MSM_SLICE);
    std::vector<i32> partials(((size_t)slices + 1) * 2 * XY, 0x5a5a5a5a);
    for (u32 lane = 0; lane < slices; lane++)
        msm_slice_lane<F, true>(table.data(), entries.data(), offsets.data(), bpw, E, lane, buckets.data(), partials.data());
    for (u32 key = 0; key < bpw; key++) msm_combine_lane<F>(offsets.data(), totals.data(), key, partials.data(), buckets.data());
    u32 seg_per_window = (bpw + MSM_SEG - 1) / MSM_SEG;
    ec_xyzz<F> total = ec_infinity<F>();
    for (u32 si = 0; si < seg_per_window; si++) total = ec_add(total, msm_segment_lane<F>(buckets.data(), pm, 0, si));
    return ec_to_affine_gnark(out, total);
}
extern "C" {
u32 hostsim_msm_fixed_g1(const u64 *points, const u64 *scalars, u32 n, u32 *out16) { return hostsim_msm_fixed<FpField>(points, scalars, n, out16); }
u32 hostsim_msm_fixed_g2(const u64 *points, const u64 *scalars, u32 n, u32 *out32) { return hostsim_msm_fixed<Fp2Field>(points, scalars, n, out32); }
u32 hostsim_msm_g1(const u64 *points, const u64 *scalars, u32 n, u32 mode, u32 heavy_min, u32 heavy_threads, u32 *out16) {
    return hostsim_msm<FpField>(points, scalars, n, mode, heavy_min, heavy_threads, out16);
}
u32 hostsim_msm_g2(const u64 *points, const u64 *scalars, u32 n, u32 mode, u32 heavy_min, u32 heavy_threads, u32 *out32) {
    return hostsim_msm<Fp2Field>(points, scalars, n, mode, heavy_min, heavy_threads, out32);
}
void hostsim_msm_glv_split(const u64 *scalar4, u32 *m1, u32 *m2, u32 *negs) {
    u32 sw[8];
    msm_load_scalar(scalar4, 0, sw);
    msm_glv_split(sw, m1, negs[0], m2, negs[1]);
}
void hostsim_msm_plan(u64 n, u32 *out8) {
    msm_plan pl = msm_make_plan(n);
    out8[0] = pl.n_pad; out8[1] = pl.c; out8[2] = pl.windows; out8[3] = pl.buckets_per_window; out8[4] = pl.total_buckets;
    out8[5] = pl.chunks; out8[6] = pl.chunk_len; out8[7] = MSM_SEG;
}
int hostsim_msm_digit_roundtrip(const u64 *scalar4, u32 c, int *digits, u32 *codes) {
    u32 sw[8], carry = 0, windows = 254 / c + 1;
    msm_load_scalar(scalar4, 0, sw);
    for (u32 w = 0; w < windows; w++) {
        digits[w] = msm_digit(sw, w, c, carry);
        codes[w] = msm_digit_code(digits[w]);
    }
    return (int)carry;
}

// Poseidon-BN254: states as 4 x 8 words, regular (non-Montgomery) form
void hostsim_poseidon_bn254_permute(u32 *st32) {
    fr s[4];
    for (int i = 0; i < 4; i++) s[i] = fr_from_regular(st32 + 8 * i);
    poseidon_bn254_permute(s);
    for (int i = 0; i < 4; i++) fr_to_regular(st32 + 8 * i, s[i]);
}
void hostsim_poseidon_bn254_permute_coop(u32 *st32) {      // the four-lane form of the permutation, lane by lane
    fr s[4];
    for (int i = 0; i < 4; i++) s[i] = fr_from_regular(st32 + 8 * i);
    poseidon_bn254_permute_coop_ref(s);
    for (int i = 0; i < 4; i++) fr_to_regular(st32 + 8 * i, s[i]);
}
void hostsim_poseidon_bn254_hash(const u64 *in, u32 len, u32 *out8) { poseidon_bn254_hash_or_noop(in, 1, len, out8); }
void hostsim_poseidon_bn254_two_to_one(const u32 *l, const u32 *r, u32 *out8) { poseidon_bn254_two_to_one(l, r, out8); }

// plonky2 gate evaluators (csrc/plonky2_gates.cuh): sum_i alpha_c^i * constraint_i of one gate at one point, and the filter
void hostsim_p2_eval_gate(u32 type, const u32 *params, const u64 *extra, const u64 *wires, u32 n_wires, const u64 *consts,
                          u32 n_consts, const u64 *pih, const u64 *alpha, u32 nch, u64 *acc_out) {
    p2_gate g;
    g.type = type;
    for (int k = 0; k < 4; k++) g.p[k] = params[k];
    g.selector_index = 0;
    g.group_start = g.group_end = 0;
    g.extra_off = 0;
    p2_vars v;
    v.wires = wires;   // stride 1, p = 0: column j at wires[j]
    v.consts = consts;
    v.stride = 1;
    v.p = 0;
    v.nsel = 0;
    for (int k = 0; k < 4; k++) v.pih[k] = pih[k];
    static u32 tab[P2_MAX_CH][1024 * 6];
    p2_consumer out;
    out.nch = (int)nch;
    for (int c = 0; c < P2_MAX_CH; c++) {
        u64 a = c < (int)nch ? alpha[c] : 0, pw = 1;
        for (int i = 0; i < 1024; i++) {
            gl_limbs22(pw, &tab[c][6 * i]);
            pw = gl_mul(pw, a);
        }
        out.apow[c] = tab[c];
    }
    out.reset(0);
    p2_eval_gate(g, v, extra, out);
    for (u32 c = 0; c < nch; c++) acc_out[c] = out.result((int)c);
    (void)n_wires;
    (void)n_consts;
}
// the opt-in whole-round / lazy-partial-round evaluator of the Poseidon gate (ZKLC_P2_POSEIDON_GATE=lazy; host forms of its pieces)
void hostsim_p2_eval_poseidon_lazy(const u64 *wires, const u64 *alpha, u32 nch, u64 *acc_out, u32 mode) {
    p2_vars v;
    v.wires = wires;
    v.consts = wires;
    v.stride = 1;
    v.p = 0;
    v.nsel = 0;
    static u32 tab[P2_MAX_CH][1024 * 6];
    p2_consumer out;
    out.nch = (int)nch;
    for (int c = 0; c < P2_MAX_CH; c++) {
        u64 a = c < (int)nch ? alpha[c] : 0, pw = 1;
        for (int i = 0; i < 1024; i++) {
            gl_limbs22(pw, &tab[c][6 * i]);
            pw = gl_mul(pw, a);
        }
        out.apow[c] = tab[c];
    }
    out.reset(0);
    if (mode == 2)
        p2_eval_poseidon_loose(v, out);
    else if (mode & 1)
        p2_eval_poseidon_lazy<p2_vars, 1>(v, out);
    else
        p2_eval_poseidon_lazy<p2_vars, 0>(v, out);
    for (u32 c = 0; c < nch; c++) acc_out[c] = out.result((int)c);
}

// the U32AddMany LDS-tile evaluator (p2_quotient_addmany_tile_kernel) walked for ONE LDE point: the four waves of the workgroup one
// after the other per phase, the tile arrays with the kernel's [column][64 lanes] layout (lane 0 used).  variants: n x (num_addends,
// num_ops); slots[4][2] = the host plan (position in the list or 0xFFFFFFFF).  out[v * nch + c] = sum_i alpha_c^i constraint_i of
// variant v, comparable with hostsim_p2_eval_gate of that variant.
void hostsim_p2_addmany_tile(const u32 *variants, u32 n, const u32 *slots, const u64 *wires, const u64 *alpha, u32 nch, u64 *out, u32 k0) {
    static u32 tab[P2_MAX_CH][1024 * 6];
    gl_ktab *apow[P2_MAX_CH];
    for (int c = 0; c < P2_MAX_CH; c++) {
        u64 a = c < (int)nch ? alpha[c] : 0, pw = 1;
        for (int i = 0; i < 1024; i++) {
            gl_limbs22(pw, &tab[c][6 * i]);
            pw = gl_mul(pw, a);
        }
        apow[c] = tab[c];
    }
    const u32 COLS = 48, WAVES = 4, per_wave = COLS / WAVES;
    u32 lo = 0xFFFFFFFFu, hi = 0;
    for (u32 v = 0; v < n; v++) {
        u32 na = variants[2 * v], ops = variants[2 * v + 1], l0 = (na + 3) * ops;
        lo = l0 < lo ? l0 : lo;
        hi = l0 + 18 * ops > hi ? l0 + 18 * ops : hi;
    }
    gl_acc3 acc[WAVES][2][P2_MAX_CH];
    u64 comb[WAVES][2];
    u32 OPS[WAVES][2], L0[WAVES][2];
    p2_vars pv;
    pv.wires = wires;
    pv.stride = 1;
    pv.p = 0;
    for (u32 w = 0; w < WAVES; w++)
        for (int s = 0; s < 2; s++) {
            for (int c = 0; c < P2_MAX_CH; c++) acc[w][s][c].c0 = acc[w][s][c].c1 = acc[w][s][c].c2 = 0;
            comb[w][s] = 0;
            OPS[w][s] = L0[w][s] = 0;
            u32 pos = slots[2 * w + s];
            if (pos == 0xFFFFFFFFu) continue;
            u32 na = variants[2 * pos], ops = variants[2 * pos + 1];
            OPS[w][s] = ops;
            L0[w][s] = (na + 3) * ops;
            if (s == 0)
                p2_amt_routed<0>(acc[w], pv, na, ops, apow, (int)nch, k0);
            else
                p2_amt_routed<1>(acc[w], pv, na, ops, apow, (int)nch, k0);
        }
    static u64 tw[48 * 64], trp[48 * 64];
    for (u32 top = hi; top > lo;) {
        const u32 base = top - lo > COLS ? top - COLS : lo;
        for (u32 w = 0; w < WAVES; w++) {
            u64 x[12], rp[12];
            const u32 first = base + w * per_wave;
            for (u32 j = 0; j < per_wave; j++) x[j] = wires[first + j < top ? first + j : top - 1];
            p2_range_products4<12>(x, rp);
            for (u32 j = 0; j < per_wave; j++)
                if (first + j < top) {
                    tw[(size_t)(first + j - base) * 64] = x[j];
                    trp[(size_t)(first + j - base) * 64] = rp[j];
                }
        }
        for (u32 w = 0; w < WAVES; w++) {
            if (OPS[w][0]) p2_amt_consume<0>(acc[w], comb[w], OPS[w][0], L0[w][0], base, top, tw, trp, 0, apow, (int)nch, k0);
            if (OPS[w][1]) p2_amt_consume<1>(acc[w], comb[w], OPS[w][1], L0[w][1], base, top, tw, trp, 0, apow, (int)nch, k0);
        }
        top = base;
    }
    for (u32 w = 0; w < WAVES; w++)
        for (int s = 0; s < 2; s++) {
            u32 pos = slots[2 * w + s];
            if (pos == 0xFFFFFFFFu) continue;
            for (u32 c = 0; c < nch; c++) out[pos * nch + c] = gl_acc3_reduce(acc[w][s][c]);
        }
}
u64 hostsim_p2_filter(u32 row, u32 start, u32 end, u64 s, u32 many) { return p2_filter(row, start, end, s, many != 0); }
void hostsim_gl2_op(int op, const u64 *a, const u64 *b, u64 e, u64 *out) {
    gl2 x = gl2_make(a[0], a[1]), y = gl2_make(b[0], b[1]), r;
    switch (op) {
        case 0: r = gl2_add(x, y); break;
        case 1: r = gl2_sub(x, y); break;
        case 2: r = gl2_mul(x, y); break;
        case 3: r = gl2_sqr(x); break;
        case 4: r = gl2_inv(x); break;
        case 5: r = gl2_pow(x, e); break;
        default: r = gl2_make(0, 0);
    }
    out[0] = r.a;
    out[1] = r.b;
}

// ---- BN254 pairing (csrc/bn254_pairing.cuh).  g1: k x 16 words (x, y gnark Montgomery), g2: k x 32 words (X.A0, X.A1, Y.A0, Y.A1);
// all-zero point = infinity (pair skipped).  stage 0: Miller loops only, 1: + final exponentiation.  gt_out: 96 words.
u32 hostsim_pairing(const u32 *g1, const u32 *g2, u32 k, u32 stage, u32 *gt_out) {
    fp12 f = f12_one();
    for (u32 i = 0; i < k; i++) {
        u32 z1 = 0, z2 = 0;
        for (int j = 0; j < 16; j++) z1 |= g1[16 * i + j];
        for (int j = 0; j < 32; j++) z2 |= g2[32 * i + j];
        if (!z1 || !z2) continue;
        fp xp = fp_reduce(fp_from_gnark(g1 + 16 * i)), yp = fp_reduce(fp_from_gnark(g1 + 16 * i + 8));
        fp2 xq = fp2_reduce(fp2_from_gnark(g2 + 32 * i)), yq = fp2_reduce(fp2_from_gnark(g2 + 32 * i + 16));
        bn_miller_loop(f, xp, yp, xq, yq);
    }
    if (stage) f = bn_final_exponentiation(f);
    f12_to_gnark(gt_out, f);
    return f12_is_one(f);
}
// tower primitives: op 0 mul, 1 sqr, 2 inv, 3 frobenius, 4 frobenius^2, 5 conj; operands/results = 96 gnark words
void hostsim_f12_op(int op, const u32 *a96, const u32 *b96, u32 *out96) {
    auto load = [](const u32 *w) {
        fp12 r;
        fp2 *x[6] = {&r.c0.b0, &r.c0.b1, &r.c0.b2, &r.c1.b0, &r.c1.b1, &r.c1.b2};
        for (int i = 0; i < 6; i++) *x[i] = fp2_reduce(fp2_from_gnark(w + 16 * i));
        return r;
    };
    fp12 a = load(a96), b = load(b96), r;
    switch (op) {
        case 0: r = f12_mul(a, b); break;
        case 1: r = f12_sqr(a); break;
        case 2: r = f12_inv(a); break;
        case 3: r = f12_frobenius(a, 1); break;
        case 4: r = f12_frobenius(a, 2); break;
        default: r = f12_conj(a);
    }
    f12_to_gnark(out96, r);
}
// weak reduction on a raw limb vector (|value| <= 64p): returns the canonical value of the result as gnark-domain words
void hostsim_fp_wred(const i32 *limbs, i32 *out_limbs, u32 *out_words) {
    fp a;
    for (int i = 0; i < 10; i++) a.v[i] = limbs[i];
    fp r = fp_wred(a);
    for (int i = 0; i < 10; i++) out_limbs[i] = r.v[i];
    fp_freeze_words(out_words, r);
}
// The two-pass Fr NTT of csrc/bn254_fr_ntt.hip walked sequentially: the table block, then pass A tile by tile (loads through the
// bit reversal, the stages butterfly by butterfly, the inter-pass twiddle), then pass B -- the same ZKLC_HD functions as the kernels.
// data: n = 2^log_n elements in gnark-crypto's layout (4 x u64 Montgomery), transformed in place.
void hostsim_fr_ntt_two_pass(u64 *data, u32 log_n, u32 inverse, u32 coset) {
    frn_plan p = frn_make_plan(log_n);
    const u32 N1 = 1u << p.t1, N2 = 1u << p.t2;
    std::vector<i32> tab((size_t)frn_table_elems(p) * 10), mid((size_t)N1 * N2 * 10), tile((size_t)N1 * 10);
    frn_table_consts(tab.data(), log_n, inverse);
    for (u32 e = 4; e < frn_table_elems(p); e++) frn_store(tab.data() + (size_t)e * 10, frn_table_entry(tab.data(), p, e));
    for (u32 j2 = 0; j2 < N2; j2++) {
        for (u32 j1 = 0; j1 < N1; j1++)
            frn_tile_store(tile.data(), N1, frn_bitrev(j1, p.t1), frn_pass_a_in(data, tab.data(), p, j1, j2, coset && !inverse));
        for (u32 t = 0; t < p.t1; t++)
            for (u32 b = 0; b < N1 / 2; b++) frn_tile_butterfly(tile.data(), p.t1, t, b, tab.data() + (size_t)frn_off_loc1(p) * 10);
        for (u32 k1 = 0; k1 < N1; k1++)
            frn_store(mid.data() + ((size_t)j2 * N1 + k1) * 10, frn_pass_a_out(frn_tile_load(tile.data(), N1, k1), tab.data(), p, k1, j2));
    }
    for (u32 k1 = 0; k1 < N1; k1++) {
        for (u32 j2 = 0; j2 < N2; j2++)
            frn_tile_store(tile.data(), N2, frn_bitrev(j2, p.t2), frn_load(mid.data() + ((size_t)j2 * N1 + k1) * 10));
        for (u32 t = 0; t < p.t2; t++)
            for (u32 b = 0; b < N2 / 2; b++) frn_tile_butterfly(tile.data(), p.t2, t, b, tab.data() + (size_t)frn_off_loc2(p) * 10);
        for (u32 k2 = 0; k2 < N2; k2++) frn_pass_b_out(data, frn_tile_load(tile.data(), N2, k2), tab.data(), p, k1, k2, coset && inverse);
    }
}
// x * 2^e (e < 96) of the NTT group's inner twiddles
u64 hostsim_gl_mul_2exp(u64 x, u32 e) { return gl_mul_2exp(x, e); }
// One butterfly group of the Goldilocks NTT passes in both forms: the product's (shift twiddles inside, one table multiplication
// per element outside) and the plain definition (every butterfly with its full twiddle).  x: 2^g values, overwritten with the
// product form's result; plain: receives the definition's.  Returns the number of elements (0 = unsupported g).
u32 hostsim_gl_ntt_group(u32 g, u32 dit, u32 inverse, u32 logn, u32 s_first, u64 J, u64 *x, u64 *plain) {
    u64 w = gl_root_of_unity(logn);
    if (inverse) w = gl_inv(w);
    const u32 M = 1u << g;
    u64 tab[16];
    for (u32 m = 1; m < M; m++) tab[m - 1] = gl_pow(w, ((u64)gl_bitrev_small(m, (int)g) * J) << s_first);
    for (u32 m = 0; m < M; m++) plain[m] = x[m];
#define GRP(G)                                                                                       \
    case G:                                                                                          \
        if (dit) {                                                                                   \
            gl_ntt_group_plain<G, true>(plain, w, logn, s_first, J);                                 \
            if (inverse) gl_ntt_group_regs<G, true, true>(x, tab); else gl_ntt_group_regs<G, true, false>(x, tab);    \
        } else {                                                                                     \
            gl_ntt_group_plain<G, false>(plain, w, logn, s_first, J);                                \
            if (inverse) gl_ntt_group_regs<G, false, true>(x, tab); else gl_ntt_group_regs<G, false, false>(x, tab);  \
        }                                                                                            \
        return M;
    switch (g) {
        GRP(1) GRP(2) GRP(3) GRP(4)
    }
#undef GRP
    return 0;
}
// The zero-aware first group of an 8x LDE (gl_ntt_group_regs<G, false, false, 3>): x holds 2^g values of which only the first
// 2^(g - 3) are taken (the others are the zero padding and are NOT read); general: the general group on the same values padded with
// zeros.  Returns the number of elements (0 = unsupported g).
u32 hostsim_gl_ntt_group_zero_padded(u32 g, u32 logn, u32 s_first, u64 J, u64 *x, u64 *general) {
    u64 w = gl_root_of_unity(logn);
    const u32 M = 1u << g;
    u64 tab[16];
    for (u32 m = 1; m < M; m++) tab[m - 1] = gl_pow(w, ((u64)gl_bitrev_small(m, (int)g) * J) << s_first);
    for (u32 m = 0; m < M; m++) general[m] = m < (M >> 3) ? x[m] : 0;
    for (u32 m = (M >> 3); m < M; m++) x[m] = 0xDEADBEEFDEADBEEFull;          // must not be read
    switch (g) {
        case 3: gl_ntt_group_regs<3, false, false>(general, tab); gl_ntt_group_regs<3, false, false, 3>(x, tab); return M;
        case 4: gl_ntt_group_regs<4, false, false>(general, tab); gl_ntt_group_regs<4, false, false, 3>(x, tab); return M;
    }
    return 0;
}
u32 hostsim_g2_op(int op, const u32 *p32, const u32 *q32, u32 *out32) {
    g2_xyzz a;
    a.X = fp2_reduce(fp2_from_gnark(p32));
    a.Y = fp2_reduce(fp2_from_gnark(p32 + 16));
    a.ZZ = a.ZZZ = fp2_one();
    fp2 qx = fp2_from_gnark(q32), qy = fp2_from_gnark(q32 + 16);
    g2_xyzz r;
    switch (op) {
        case 0: r = ec_add_affine<Fp2Field>(a, qx, qy, 0); break;
        case 1: r = ec_add_affine<Fp2Field>(a, qx, qy, 1); break;
        case 2: r = ec_double(a); break;
        default: { g2_xyzz s = ec_add_affine<Fp2Field>(a, qx, qy, 0); r = ec_add(s, a); }
    }
    return ec_to_affine_gnark(out32, r);
}
}
