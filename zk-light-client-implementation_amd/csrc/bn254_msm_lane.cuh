// Lane-level pieces of the BN254 multi-scalar multiplication (bn254_msm.hip): everything ONE lane does -- scalar recoding,
// the bucket accumulation loop, the segment running sums -- as ZKLC_HD functions over plain pointers, so that tests/hostsim
// runs the very same code on the CPU (a sequential walk over the lanes) against the oracle's naive multi-exponentiation.
// Replaces gnark-crypto's `G1Affine.MultiExp` / `G2Affine.MultiExp` inner loops (un-vendored, gnark-plonky2-verifier/go.mod:9;
// call site `groth16.Prove`, gnark-plonky2-verifier/cmd/web-api.go:77).
#pragma once
#include "bn254_ec.cuh"

#define MSM_MAX_WINDOWS 32
#define MSM_HEAVY 4096u      // buckets with more entries go to the workgroup-per-bucket kernel
#define MSM_MAX_HEAVY 2048u
#define MSM_SEG 16u          // buckets per running-sum segment (one lane each)

struct msm_plan {
    u32 n, n_pad;            // points; digit-row stride (n rounded up to 8)
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

// acc += the `cnt` points listed in entries[beg ..) (entry = point index << 1 | negate), `step` apart (1: a lane owns the bucket;
// the workgroup size in the heavy-bucket kernel).  Software-pipelined: the raw words of the NEXT point and the entry after it are
// requested before the current addition starts, so the two dependent gathers (entry -> point) of an iteration overlap ~3 000
// instructions of field arithmetic instead of stalling every wave of the SIMD at the same place.
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

MSM_HOST_DEV msm_plan msm_make_plan(u64 n) {
    msm_plan pl;
    pl.n = (u32)n;
    pl.n_pad = (u32)((n + 7) & ~(u64)7);
    pl.c = msm_pick_window(n);
    pl.windows = 254 / pl.c + 1;
    pl.buckets_per_window = 1u << (pl.c - 1);
    pl.total_buckets = pl.windows * pl.buckets_per_window;
    // (chunk, window) tiles of the counting sort: enough tiles to fill the chip, chunks of at least 2^13 points
    u32 chunks = (u32)(n >> 13);
    if (chunks < 1) chunks = 1;
    if (chunks > 16) chunks = 16;
    pl.chunks = chunks;
    pl.chunk_len = ((pl.n_pad / 8 + chunks - 1) / chunks) * 8;
    if (pl.chunk_len == 0) pl.chunk_len = 8;
    return pl;
}
