// Lane-level pieces of the BN254 multi-scalar multiplication (bn254_msm.hip): everything ONE lane does -- scalar recoding,
// the bucket accumulation loop, the segment running sums -- as ZKLC_HD functions over plain pointers, so that tests/hostsim
// runs the very same code on the CPU (a sequential walk over the lanes) against the oracle's naive multi-exponentiation.
// Replaces gnark-crypto's `G1Affine.MultiExp` / `G2Affine.MultiExp` inner loops (un-vendored, gnark-plonky2-verifier/go.mod:9;
// call site `groth16.Prove`, gnark-plonky2-verifier/cmd/web-api.go:77).
#pragma once
#include "bn254_ec.cuh"

#define MSM_MAX_WINDOWS 32
#define MSM_MAX_HEAVY 2048u  // buckets cut into more than MSM_COMBINE_SERIAL slices (skewed scalars): summed by a workgroup each
#define MSM_SEG 16u          // buckets per running-sum segment (one lane each)

struct msm_plan {
    u32 n, n_pad;            // ITEMS (= points, or 2 x points with the endomorphism split); digit-row stride (n rounded up to 8)
    u32 n_pts, glv;          // input points; 1 = every point enters twice: (k1, P) and (k2, phi(P)) with 127-bit scalars (G1 only)
    u32 c, windows, buckets_per_window, total_buckets;
    u32 chunks, chunk_len;   // the counting sort works on (chunk, window) tiles; chunk_len is a multiple of 8
};

// 8 little-endian words of scalar i, reduced below r: a scalar that is not reduced (anything up to 2^256 - 1) would lose its
// top bits in the 254-bit window recoding; the points have order r, so the sum is the same
ZKLC_HD void msm_load_scalar(const u64 *scalars, u32 i, u32 *w) {
    const u64 *p = scalars + (size_t)i * 4;
    u64 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
    w[0] = (u32)a0; w[1] = (u32)(a0 >> 32); w[2] = (u32)a1; w[3] = (u32)(a1 >> 32);
    w[4] = (u32)a2; w[5] = (u32)(a2 >> 32); w[6] = (u32)a3; w[7] = (u32)(a3 >> 32);
    const u32 R[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int it = 0; it < 6; it++) {       // 2^256 / r < 6
        bool ge = true;
        for (int k = 7; k >= 0; k--)
            if (w[k] != R[k]) {
                ge = w[k] > R[k];
                break;
            }
        if (!ge) break;
        u64 borrow = 0;
        for (int k = 0; k < 8; k++) {
            u64 d = (u64)w[k] - R[k] - borrow;
            w[k] = (u32)d;
            borrow = (d >> 32) & 1;
        }
    }
}

// signed digit of window w given the running carry (updated): digit in [-(2^(c-1) - 1), 2^(c-1)]
ZKLC_HD int msm_digit(const u32 *sw, u32 w, u32 c, u32 &carry) {
    u32 bit = w * c, wi = bit >> 5, sh = bit & 31;
    u64 x = (u64)sw[wi] >> sh;
    if (wi + 1 < 8) x |= (u64)sw[wi + 1] << (32 - sh);
    u32 raw = ((u32)x & ((1u << c) - 1)) + carry;
    if (raw > (1u << (c - 1))) {
        carry = 1;
        return (int)raw - (int)(1u << c);
    }
    carry = 0;
    return (int)raw;
}

// a digit as 16 bits (c <= 16): 0 = nothing to add; 0x8000 = +2^15 (the one value a signed 16-bit number cannot hold: the range
// is [-(2^15 - 1), 2^15]); anything else is the two's complement digit
ZKLC_HD u32 msm_digit_code(int d) { return (u32)d & 0xffffu; }
// -> bucket index |d| - 1 and the sign; call only for code != 0
ZKLC_HD u32 msm_code_bucket(u32 code, u32 &neg) {
    neg = (code > 0x8000u) ? 1u : 0u;
    u32 mag = neg ? 0x10000u - code : code;
    return mag - 1;
}

// ---- the endomorphism split (G1, round 4): BN254 has phi(x, y) = (beta x, y) = lambda (x, y) with beta^3 = 1 in Fp and
// lambda^2 + lambda + 1 = 0 in Fr, so  k P = k1 P + k2 phi(P)  with |k1|, |k2| < 2^127 (Gallant-Lambert-Vanstone): the same number
// of bucket additions (2 n items x 8 windows of 16 bits instead of n x 16) but HALF the windows -- half the bucket reductions and
// 112 instead of 240 serial doublings at the end.  (v1, v2) = ((a1, b1), (a2, b2)) is a short basis of the lattice
// {(x, y): x + y lambda = 0 mod r} (extended Euclid on (r, lambda)); c_i = round(k g_i / 2^256) with g1 = round(2^256 b2 / r),
// g2 = round(-2^256 b1 / r);  k1 = k - c1 a1 - c2 a2,  k2 = -c1 b1 - c2 b2.  k1 + k2 lambda = k (mod r) holds for ANY integers c1, c2
// (the rounding only bounds the sizes: |k_i| <= (|v1| + |v2|) / 2 (1 + eps) < 2^126.1), which is why approximate quotients are exact.
// Constants checked by tests/test_hostsim_msm.py against the oracle's curve arithmetic.
#define MSM_GLV_MIN_POINTS 256u
#define MSM_GLV_MAX_POINTS (1u << 21)   // measured (profiles/r04l_*): 2^16 -11 %, 2^18 -14 %, 2^20 -6 %, 2^22 +-0 (the doubled record table costs
                                        // the gathers of the slice kernel what the halved tail saves): no split above 2^21 points
ZKLC_HD void msm_mp_muladd(u32 *out, int nout, const u32 *a, int na, const u32 *b, int nb) {      // out += a * b (mod 2^(32 nout))
    for (int i = 0; i < na; i++) {
        u64 carry = 0;
        for (int j = 0; j < nb && i + j < nout; j++) {
            u64 t = (u64)a[i] * b[j] + out[i + j] + carry;
            out[i + j] = (u32)t;
            carry = t >> 32;
        }
        for (int k = i + nb; k < nout && carry; k++) {
            u64 t = (u64)out[k] + carry;
            out[k] = (u32)t;
            carry = t >> 32;
        }
    }
}
// 256-bit two's complement v -> (|v| as 4 words, sign); |v| < 2^127 by the bound above
ZKLC_HD void msm_glv_magnitude(const u32 *v8, u32 *m4, u32 &neg) {
    neg = v8[7] >> 31;
    u64 carry = neg;
    for (int i = 0; i < 4; i++) {
        u64 t = (u64)(neg ? ~v8[i] : v8[i]) + carry;
        m4[i] = (u32)t;
        carry = t >> 32;
    }
}
// sw: 8 words of a scalar < r  ->  magnitudes (4 words each) and signs of k1, k2
ZKLC_HD void msm_glv_split(const u32 *sw, u32 *m1, u32 &neg1, u32 *m2, u32 &neg2) {
    const u32 G1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x2u}, G2[5] = {0x391eb18eu, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x2u};
    const u32 A1[2] = {0x94d213e3u, 0x89d32568u}, A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
    const u32 NB1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u}, B2[2] = {0x94d213e3u, 0x89d32568u};
    u32 t[13], c1[3], c2[5];
    for (int i = 0; i < 13; i++) t[i] = 0;
    t[7] = 0x80000000u;                                 // + 2^255: round to nearest
    msm_mp_muladd(t, 11, sw, 8, G1, 3);
    for (int i = 0; i < 3; i++) c1[i] = t[8 + i];
    for (int i = 0; i < 13; i++) t[i] = 0;
    t[7] = 0x80000000u;
    msm_mp_muladd(t, 13, sw, 8, G2, 5);
    for (int i = 0; i < 5; i++) c2[i] = t[8 + i];
    // k1 = k - (c1 a1 + c2 a2),  k2 = c1 |b1| - c2 b2   (mod 2^256; the true values are small signed numbers)
    u32 s[8], k1[8], k2[8], u[8];
    for (int i = 0; i < 8; i++) s[i] = 0;
    msm_mp_muladd(s, 8, c1, 3, A1, 2);
    msm_mp_muladd(s, 8, c2, 5, A2, 4);
    u64 borrow = 0;
    for (int i = 0; i < 8; i++) {
        u64 d = (u64)sw[i] - s[i] - borrow;
        k1[i] = (u32)d;
        borrow = (d >> 32) & 1;
    }
    for (int i = 0; i < 8; i++) s[i] = u[i] = 0;
    msm_mp_muladd(s, 8, c1, 3, NB1, 4);
    msm_mp_muladd(u, 8, c2, 5, B2, 2);
    borrow = 0;
    for (int i = 0; i < 8; i++) {
        u64 d = (u64)s[i] - u[i] - borrow;
        k2[i] = (u32)d;
        borrow = (d >> 32) & 1;
    }
    msm_glv_magnitude(k1, m1, neg1);
    msm_glv_magnitude(k2, m2, neg2);
}

template <int AFF>  // u64 words per affine point: 8 (G1) or 16 (G2)
ZKLC_HD bool msm_point_is_inf(const u64 *points, u32 i) {
    const u64 *p = points + (size_t)i * AFF;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < AFF; k++) acc |= p[k];
    return acc == 0;  // gnark encodes infinity as all-zero coordinates
}

template <class F>
struct msm_cfg {
    static constexpr int XYZZ = 4 * F::LIMBS;            // i32 words of a stored XYZZ point
    static constexpr int AFF = 2 * 4 * F::LIMBS / 10;    // u64 words of an affine point at the ABI
    static constexpr int BLOCK = F::LIMBS == 10 ? 256 : 128;  // workgroup size of the LDS tree reductions (<= 40 KiB of LDS)
};

// the raw words of an affine point: fetched one iteration ahead of their use in the bucket loop
template <class F>
struct msm_raw_point {
    u64 q[msm_cfg<F>::AFF];
};
template <class F>
ZKLC_HD void msm_fetch_raw(msm_raw_point<F> &r, const u64 *points, u32 idx) {
    const int W = msm_cfg<F>::AFF;
#if defined(__HIPCC__)
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(points + (size_t)idx * W);
#pragma unroll
    for (int k = 0; k < W / 2; k++) {
        ulonglong2 a = p[k];
        r.q[2 * k] = a.x;
        r.q[2 * k + 1] = a.y;
    }
#else
    for (int k = 0; k < W; k++) r.q[k] = points[(size_t)idx * W + k];
#endif
}
template <class F>
ZKLC_HD void msm_raw_to_xy(const msm_raw_point<F> &r, typename F::T &x, typename F::T &y) {
    const int W = msm_cfg<F>::AFF;  // u32 words per coordinate
    u32 w[2 * W];
#pragma unroll
    for (int k = 0; k < W; k++) {
        w[2 * k] = (u32)r.q[k];
        w[2 * k + 1] = (u32)(r.q[k] >> 32);
    }
    x = F::from_gnark(w);
    y = F::from_gnark(w + W);
}
template <class F>
ZKLC_HD void msm_load_point(const u64 *points, u32 idx, typename F::T &x, typename F::T &y) {
    msm_raw_point<F> r;
    msm_fetch_raw<F>(r, points, idx);
    msm_raw_to_xy<F>(r, x, y);
}

template <class F>
ZKLC_HD void msm_store_xyzz(i32 *dst, const ec_xyzz<F> &p) {
    F::store(dst, p.X);
    F::store(dst + F::LIMBS, p.Y);
    F::store(dst + 2 * F::LIMBS, p.ZZ);
    F::store(dst + 3 * F::LIMBS, p.ZZZ);
}
template <class F>
ZKLC_HD ec_xyzz<F> msm_load_xyzz(const i32 *src) {
    ec_xyzz<F> p;
    p.X = F::load(src);
    p.Y = F::load(src + F::LIMBS);
    p.ZZ = F::load(src + 2 * F::LIMBS);
    p.ZZZ = F::load(src + 3 * F::LIMBS);
    return p;
}

// acc += the `cnt` points listed in entries[beg ..) (entry = point index << 1 | negate), `step` apart.  Software-pipelined: the raw
// words of the NEXT point and the entry after it are requested before the current addition starts.  (The strided form is what a
// workgroup of the heavy-combine kernel would do on raw entries; the product path is msm_slice_lane below.)
template <class F>
ZKLC_HD void msm_bucket_lane(ec_xyzz<F> &acc, const u64 *points, const u32 *entries, u32 beg, u32 first, u32 cnt, u32 step) {
    if (first >= cnt) return;
    u32 ent_cur = entries[beg + first];
    msm_raw_point<F> raw_cur;
    msm_fetch_raw<F>(raw_cur, points, ent_cur >> 1);
    u32 e_next = first + step;
    u32 ent_next = e_next < cnt ? entries[beg + e_next] : 0;
    for (u32 e = first; e < cnt; e += step) {
        msm_raw_point<F> raw_next = raw_cur;
        u32 ent_next2 = 0;
        if (e + step < cnt) {
            msm_fetch_raw<F>(raw_next, points, ent_next >> 1);
            if (e + 2 * step < cnt) ent_next2 = entries[beg + e + 2 * step];
        }
        typename F::T x, y;
        msm_raw_to_xy<F>(raw_cur, x, y);
        acc = ec_add_affine<F>(acc, x, y, ent_cur & 1);
        raw_cur = raw_next;
        ent_cur = ent_next;
        ent_next = ent_next2;
    }
}

// ---- the product path of the bucket accumulation: SLICES of the sorted entry list
// Lane L owns entries [L * MSM_SLICE, (L + 1) * MSM_SLICE) of the list sorted by (window, bucket), whatever buckets they belong to:
// every lane performs exactly MSM_SLICE additions, so neither the Poisson spread of the bucket sizes (+22 % per wave with one lane
// per bucket) nor the skew of the top window (254 = 15 * 16 + 14: its buckets are 4-8 x heavier and used to run as a long tail on
// a few waves while the chip was idle -- round 3, profiles/r03f) costs anything.  Inside its slice a lane walks bucket segments:
// a segment that is a WHOLE bucket is stored to the bucket array, the (at most two) segments cut by the slice boundaries go to
// partials[2 * lane + (not the lane's first segment)]; msm_combine_lane adds the partials of a bucket.  Points are read in the
// converted form (msm_convert_point: reduced internal limbs, x then y), so a segment starts by COPYING its first point -- chosen by
// a flag, no branch: the generic addition is computed and discarded -- and the loop has no infinity case.
#define MSM_SLICE 128u
#define MSM_COMBINE_SERIAL 32u    // buckets cut into more slices than this are summed by a workgroup (heavy-combine kernel)

// PK = packed records (round 4): x then y as canonical values in 8 x 32 bits per Fp coordinate -- 64 bytes per G1 point (one
// aligned half cache line per gather instead of an 80-byte record straddling two 128-byte lines), 128 per G2 point; unpacking is
// ~60 shift / mask instructions per point against ~2 800 of the addition
template <class F, bool PK>
struct msm_rec {
    static constexpr int WORDS = PK ? 2 * F::PACKW : 2 * F::LIMBS;     // 32-bit words per converted point
};
template <class F, bool PK>
ZKLC_HD void msm_convert_point(i32 *dst, const u64 *points, u32 idx) {
    typename F::T x, y;
    msm_load_point<F>(points, idx, x, y);
    if (PK) {
        F::pack(reinterpret_cast<u32 *>(dst), F::reduce(x));
        F::pack(reinterpret_cast<u32 *>(dst) + F::PACKW, F::reduce(y));
    } else {
        F::store(dst, F::reduce(x));
        F::store(dst + F::LIMBS, F::reduce(y));
    }
}

// the two records of point idx under the endomorphism split: item idx = (x, +-y), item n_pts + idx = (beta x, +-y); the signs of
// k1 / k2 are folded into y, so the digits downstream are those of the magnitudes
template <class F, bool PK>
ZKLC_HD void msm_convert_point_glv(i32 *dst1, i32 *dst2, const u64 *points, u32 idx, u32 neg1, u32 neg2) {
    typename F::T x, y;
    msm_load_point<F>(points, idx, x, y);
    x = F::reduce(x);
    y = F::reduce(y);
    const typename F::T bx = F::mul(x, F::glv_beta()), ny = F::reduce(F::neg(y));
    const typename F::T y1 = F::select(y, ny, neg1), y2 = F::select(y, ny, neg2);
    if (PK) {
        F::pack(reinterpret_cast<u32 *>(dst1), x);
        F::pack(reinterpret_cast<u32 *>(dst1) + F::PACKW, y1);
        F::pack(reinterpret_cast<u32 *>(dst2), bx);
        F::pack(reinterpret_cast<u32 *>(dst2) + F::PACKW, y2);
    } else {
        F::store(dst1, x);
        F::store(dst1 + F::LIMBS, y1);
        F::store(dst2, bx);
        F::store(dst2 + F::LIMBS, y2);
    }
}

template <class F>
struct msm_cpoint {
    i32 w[2 * F::LIMBS];
};
template <class F, bool PK>
ZKLC_HD void msm_fetch_cpoint(msm_cpoint<F> &r, const i32 *cpoints, u32 idx) {
    const int W = msm_rec<F, PK>::WORDS;
#if defined(__HIPCC__)
    const int4 *p = reinterpret_cast<const int4 *>(cpoints + (size_t)idx * W);
#pragma unroll
    for (int k = 0; k < W / 4; k++) {
        int4 a = p[k];
        r.w[4 * k] = a.x;
        r.w[4 * k + 1] = a.y;
        r.w[4 * k + 2] = a.z;
        r.w[4 * k + 3] = a.w;
    }
#else
    for (int k = 0; k < W; k++) r.w[k] = cpoints[(size_t)idx * W + k];
#endif
}
template <class F, bool PK>
ZKLC_HD void msm_cpoint_xy(const msm_cpoint<F> &r, typename F::T &x, typename F::T &y) {
    if (PK) {
        x = F::unpack(reinterpret_cast<const u32 *>(r.w));
        y = F::unpack(reinterpret_cast<const u32 *>(r.w) + F::PACKW);
    } else {
        x = F::load(r.w);
        y = F::load(r.w + F::LIMBS);
    }
}

template <class F>
ZKLC_HD ec_xyzz<F> msm_select_xyzz(const ec_xyzz<F> &a, const ec_xyzz<F> &b, u32 take_b) {
    ec_xyzz<F> r;
    r.X = F::select(a.X, b.X, take_b);
    r.Y = F::select(a.Y, b.Y, take_b);
    r.ZZ = F::select(a.ZZ, b.ZZ, take_b);
    r.ZZZ = F::select(a.ZZZ, b.ZZZ, take_b);
    return r;
}

// the exceptional additions of the slice loop (P = +-Q: adversarial inputs only; an accumulator at infinity after P + (-P)), out of
// line, taking the RECORD ADDRESS instead of the operands: by value the 2 x LIMBS words went through the stack and the compiler
// hoisted those stores above the rare branch -- scratch stores on EVERY addition (2.8 GB of writes per 2^22 multi-exponentiation)
template <class F, bool PK>
#if defined(__HIPCC__)
__device__ __attribute__((noinline))
#else
static
#endif
ec_xyzz<F> msm_add_special(const i32 *cpoints, u32 ent, u32 p_inf, u32 same_y) {
    msm_cpoint<F> raw;
    msm_fetch_cpoint<F, PK>(raw, cpoints, ent >> 1);
    typename F::T x, y;
    msm_cpoint_xy<F, PK>(raw, x, y);
    y = F::select(y, F::neg(y), ent & 1);
    ec_xyzz<F> r;
    if (p_inf) {
        r.X = F::reduce(x);
        r.Y = F::reduce(y);
        r.ZZ = F::one();
        r.ZZZ = F::one();
    } else if (same_y) {
        r = ec_double_affine<F>(F::reduce(x), F::reduce(y));
    } else {
        r = ec_infinity<F>();
    }
    return r;
}

// offsets[0 .. T): exclusive scan of the bucket sizes; E = number of entries.  One call = one lane.
template <class F, bool PK>
ZKLC_HD void msm_slice_lane(const i32 *cpoints, const u32 *entries, const u32 *offsets, u32 T, u32 E, u32 lane, i32 *buckets,
                            i32 *partials) {
    const int XY = msm_cfg<F>::XYZZ;
    const u32 lo = lane * MSM_SLICE;
    if (lo >= E) return;
    const u32 hi = lo + MSM_SLICE < E ? lo + MSM_SLICE : E;
    // the bucket holding entry lo: the largest key with offsets[key] <= lo (its successor starts beyond lo, so it is not empty)
    u32 a = 0, b = T;
    while (b - a > 1) {
        u32 m = (a + b) >> 1;
        if (offsets[m] <= lo) a = m; else b = m;
    }
    u32 key = a;
    u32 seg_start_in_slice = offsets[key] >= lo;      // the first segment starts a bucket only if the bucket starts at lo
    u32 end = key + 1 < T ? offsets[key + 1] : E;
    u32 seg_end = end < hi ? end : hi;
    u32 first_seg = 1, fresh = 1;
    ec_xyzz<F> acc;
    acc.X = acc.Y = acc.ZZ = acc.ZZZ = F::one();      // any finite value: discarded by the first (fresh) step
    u32 ent_cur = entries[lo];
    msm_cpoint<F> raw_cur;
    msm_fetch_cpoint<F, PK>(raw_cur, cpoints, ent_cur >> 1);
    u32 ent_next = lo + 1 < hi ? entries[lo + 1] : 0;
    for (u32 e = lo; e < hi; e++) {
        msm_cpoint<F> raw_next = raw_cur;
        u32 ent_next2 = 0;
        if (e + 1 < hi) {
            msm_fetch_cpoint<F, PK>(raw_next, cpoints, ent_next >> 1);
            if (e + 2 < hi) ent_next2 = entries[e + 2];
        }
        typename F::T x, y;
        msm_cpoint_xy<F, PK>(raw_cur, x, y);
        y = F::select(y, F::neg(y), ent_cur & 1);
        ec_xyzz<F> started;                            // the segment's first point as an accumulator
        started.X = x;
        started.Y = y;
        started.ZZ = started.ZZZ = F::one();
        u32 p_inf, special, same_y;
        ec_xyzz<F> sum = ec_madd_core<F>(acc, x, y, p_inf, special, same_y);
        if (special & (fresh ^ 1)) sum = msm_add_special<F, PK>(cpoints, ent_cur, p_inf, same_y);
        acc = msm_select_xyzz<F>(sum, started, fresh);
        fresh = 0;
        if (e + 1 == seg_end) {                        // segment complete (a few lanes of a wave per iteration)
            u32 whole = seg_start_in_slice & (seg_end == end);
            i32 *dst = whole ? buckets + (size_t)key * XY : partials + ((size_t)2 * lane + (first_seg ^ 1)) * XY;
            msm_store_xyzz<F>(dst, acc);
            first_seg = 0;
            fresh = 1;
            seg_start_in_slice = 1;
            if (e + 1 < hi) {
                do {                                   // next non-empty bucket (it exists: entry e + 1 belongs to one)
                    key++;
                    end = key + 1 < T ? offsets[key + 1] : E;
                } while (end <= e + 1);
                seg_end = end < hi ? end : hi;
            }
        }
        raw_cur = raw_next;
        ent_cur = ent_next;
        ent_next = ent_next2;
    }
}

// bucket `key`: nothing to do when it lies inside one slice (stored whole) or is empty (the bucket array starts zeroed = infinity);
// otherwise the sum of its partials.  Returns 1 if the bucket is cut into more than MSM_COMBINE_SERIAL slices (left to the
// heavy-combine kernel), 0 otherwise.
template <class F>
ZKLC_HD u32 msm_combine_span(const u32 *offsets, const u32 *counts, u32 key, u32 &la, u32 &lb, u32 &first_which) {
    u32 cnt = counts[key];
    if (cnt == 0) return 0;
    u32 b0 = offsets[key], end = b0 + cnt;
    la = b0 / MSM_SLICE;
    lb = (end - 1) / MSM_SLICE;
    first_which = b0 != la * MSM_SLICE;               // not the first segment of slice la unless the bucket starts the slice
    return lb > la;
}
template <class F>
ZKLC_HD void msm_combine_lane(const u32 *offsets, const u32 *counts, u32 key, const i32 *partials, i32 *buckets) {
    const int XY = msm_cfg<F>::XYZZ;
    u32 la, lb, which;
    if (!msm_combine_span<F>(offsets, counts, key, la, lb, which)) return;
    ec_xyzz<F> acc = msm_load_xyzz<F>(partials + ((size_t)2 * la + which) * XY);
    for (u32 j = la + 1; j <= lb; j++) acc = ec_add(acc, msm_load_xyzz<F>(partials + (size_t)2 * j * XY));
    msm_store_xyzz<F>(buckets + (size_t)key * XY, acc);
}

// ---- a doubling spread over the FOUR lanes of a quad (the serial tail of the MSM: 2^(c w) * window_w is c w dependent doublings,
// 240 of them for the top window -- 1.7 of the 10.8 ms of a 2^22 multi-exponentiation when one lane does each of them).
// dbl-2008-s-1 has nine products in three dependent layers: {U^2, X^2}, {U V, X V, V ZZ, M^2}, {W ZZZ, W Y, M (S - X3)}; lane `role`
// of the quad computes product `role` of each layer (operands picked by selects: one multiplication per lane and layer), the quad
// broadcasts the results (DPP on the device) and every lane keeps the whole point.  Three multiplication-times per doubling
// instead of nine.  These stage functions are shared with tests/hostsim, which walks the four lanes.
template <class F>
ZKLC_HD typename F::T ecq_select4(const typename F::T &a, const typename F::T &b, const typename F::T &c, const typename F::T &d,
                                  u32 role) {
    return F::select(F::select(a, b, role & 1), F::select(c, d, role & 1), role >> 1);
}
template <class F>
ZKLC_HD typename F::T ecq_stage1(const ec_xyzz<F> &p, u32 role) {                   // lanes 0, 2: V = U^2; lanes 1, 3: XX = X^2
    return F::sqr(F::select(F::dbl(p.Y), p.X, role & 1));
}
template <class F>
ZKLC_HD typename F::T ecq_stage2(const ec_xyzz<F> &p, const typename F::T &V, const typename F::T &XX, u32 role) {
    typedef typename F::T T;
    T U = F::dbl(p.Y), M = F::add(F::dbl(XX), XX);                                  // 3 X^2 (a = 0)
    return F::mul(ecq_select4<F>(U, p.X, V, M, role), ecq_select4<F>(V, V, p.ZZ, M, role));   // W, S, ZZ3, M^2
}
template <class F>
ZKLC_HD typename F::T ecq_stage3(const ec_xyzz<F> &p, const typename F::T &W, const typename F::T &S, const typename F::T &M,
                                 const typename F::T &X3, u32 role) {
    return F::mul(ecq_select4<F>(W, W, M, W, role), ecq_select4<F>(p.ZZZ, p.Y, F::sub(S, X3), p.ZZZ, role));   // ZZZ3, W Y, M (S - X3)
}
// the same doubling on an explicit array of the four lanes' copies of the point (the CPU check of the stage functions)
template <class F>
ZKLC_HD void ec_double_quad_ref(ec_xyzz<F> *p4) {
    typedef typename F::T T;
    T r1[4], r2[4], r3[4];
    for (u32 q = 0; q < 4; q++) r1[q] = ecq_stage1<F>(p4[q], q);
    for (u32 q = 0; q < 4; q++) r2[q] = ecq_stage2<F>(p4[q], r1[0], r1[1], q);
    T M = F::add(F::dbl(r1[1]), r1[1]);
    T X3 = F::sub(r2[3], F::dbl(r2[1]));
    for (u32 q = 0; q < 4; q++) r3[q] = ecq_stage3<F>(p4[q], r2[0], r2[1], M, X3, q);
    for (u32 q = 0; q < 4; q++) {
        p4[q].X = X3;
        p4[q].Y = F::sub(r3[2], r3[1]);
        p4[q].ZZ = r2[2];
        p4[q].ZZZ = r3[0];
    }
}

// k * p for a small k (k < 2^31), double-and-add
template <class F>
ZKLC_HD ec_xyzz<F> msm_small_mul(const ec_xyzz<F> &p, u32 k) {
    ec_xyzz<F> r = ec_infinity<F>();
    if (k == 0) return r;
    int top = 31;
    while (!((k >> top) & 1)) top--;
    for (int b = top; b >= 0; b--) {
        r = ec_double(r);
        if ((k >> b) & 1) r = ec_add(r, p);
    }
    return r;
}

// segment s of MSM_SEG buckets of window w: sum_{b in seg} (b + 1) B_b  (b = bucket index within the window) by the running-sum
// trick + one small multiple for the segment's offset
template <class F>
ZKLC_HD ec_xyzz<F> msm_segment_lane(const i32 *buckets, const msm_plan &pl, u32 w, u32 si) {
    const int XY = msm_cfg<F>::XYZZ;
    u32 lo = si * MSM_SEG, hi = lo + MSM_SEG < pl.buckets_per_window ? lo + MSM_SEG : pl.buckets_per_window;
    ec_xyzz<F> S = ec_infinity<F>(), T = ec_infinity<F>();
    for (u32 b = hi; b-- > lo;) {
        S = ec_add(S, msm_load_xyzz<F>(buckets + ((size_t)w * pl.buckets_per_window + b) * XY));
        T = ec_add(T, S);
    }
    // T = sum (b - lo + 1) B_b ; add lo * S
    if (lo) T = ec_add(T, msm_small_mul<F>(S, lo));
    return T;
}

#if defined(__HIPCC__)
#define MSM_HOST_DEV __host__ __device__ inline
#else
#define MSM_HOST_DEV static inline
#endif
MSM_HOST_DEV u32 msm_pick_window(u64 n) {
    if (n >= (1u << 19)) return 16;
    if (n >= (1u << 15)) return 14;
    if (n >= (1u << 11)) return 11;
    if (n >= 64) return 8;
    return 4;
}

MSM_HOST_DEV msm_plan msm_make_plan(u64 n_pts, bool g1 = false) {
    msm_plan pl;
    pl.n_pts = (u32)n_pts;
    pl.glv = (g1 && n_pts >= MSM_GLV_MIN_POINTS && n_pts <= MSM_GLV_MAX_POINTS) ? 1u : 0u;
    const u64 n = pl.glv ? 2 * n_pts : n_pts;
    pl.n = (u32)n;
    pl.n_pad = (u32)((n + 7) & ~(u64)7);
    pl.c = msm_pick_window(n);
    // values below 2^B need W windows with c W >= B + 1 (the top window then never carries out): B = 254, or 127 after the split
    pl.windows = pl.glv ? (127 / pl.c + 1) : (254 / pl.c + 1);
    pl.buckets_per_window = 1u << (pl.c - 1);
    pl.total_buckets = pl.windows * pl.buckets_per_window;
    // (chunk, window) tiles of the counting sort: enough tiles to fill the chip, chunks of at least 2^13 points
    u32 chunks = (u32)(n >> 13);
    if (chunks < 1) chunks = 1;
    if (chunks > 16) chunks = 16;
    if (pl.glv && chunks == 16 && n >= (1u << 18)) chunks = 32;       // half the windows: keep (chunks x windows) >= 256 tiles
    pl.chunks = chunks;
    pl.chunk_len = ((pl.n_pad / 8 + chunks - 1) / chunks) * 8;
    if (pl.chunk_len == 0) pl.chunk_len = 8;
    return pl;
}
