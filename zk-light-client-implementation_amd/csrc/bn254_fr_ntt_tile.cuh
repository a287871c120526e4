// The two-pass ("four-step") NTT over BN254 Fr for 2^12 <= n <= 2^22: what one tile / one butterfly / one table entry computes, as
// ZKLC_HD functions shared by the kernels of bn254_fr_ntt.hip and by tests/hostsim (which walks tiles and stages sequentially
// against the oracle's definition of the transform).  Replaces gnark-crypto's `fft.Domain.FFT / FFTInverse` on ecc/bn254/fr
// (un-vendored, gnark-plonky2-verifier/go.mod:9) under `groth16.Prove` (gnark-plonky2-verifier/cmd/web-api.go:77, `computeH`).
//
// n = N1 N2 (N1 = 2^t1 >= N2 = 2^t2, both <= 2^11: an N1-point tile in the ten-limb form is 80 KiB of LDS), input index
// j = j1 N2 + j2, output index k = k1 + N1 k2:
//     X[k1 + N1 k2] = sum_{j2} [ w^(j2 k1) * sum_{j1} x[j1 N2 + j2] w_N1^(j1 k1) ] w_N2^(j2 k2)
//   pass A, one workgroup per j2: the N1-point transform of the strided column x[. N2 + j2] (loaded through the bit reversal,
//           t1 radix-2 stages in LDS), times the twiddle w^(j2 k1), written CONTIGUOUSLY as T[j2][k1] in limb form;
//   pass B, one workgroup per k1: the N2-point transform of the column T[.][k1], scaled and written as X[k1 + N1 k2].
// Every element is multiplied ~15 times in all (the per-stage launches of round 1 did 2 per stage = 44, plus a 28-step power
// per element for the coset shift and again for the twiddle table on EVERY call): convert / coset shift (1-2), t1 / 2 + t2 / 2
// butterfly products -- the additions stay lazy: 11 stages grow a value to at most 14 m < 16 m, the next product reduces it --
// the inter-pass twiddle (2: w^(j2 k1) = HI[e >> 11] LO[e & 2047]) and the output scaling (1-2).  Tables are built once per
// (context, log n, direction) and stay resident.
#pragma once
#include "bn254_fr.cuh"

#define FRN_SPLIT 11u
#define FRN_SPLIT_N (1u << FRN_SPLIT)
#define FRN_FAST_MIN_LOG 12u
#define FRN_FAST_MAX_LOG 22u
#define FR_ROOT28_WORDS {0x725b19f0u, 0x9bd61b6eu, 0x41112ed4u, 0x402d111eu, 0x8ef62abcu, 0x00e0a7ebu, 0xa58a7e85u, 0x2a3c09f0u}

struct frn_plan {
    u32 log_n, t1, t2;       // N1 = 2^t1 (pass A tiles), N2 = 2^t2 (pass B tiles)
};
#if defined(__HIPCC__)
#define FRN_HOST_DEV __host__ __device__ inline
#else
#define FRN_HOST_DEV static inline
#endif
FRN_HOST_DEV frn_plan frn_make_plan(u32 log_n) {
    frn_plan p;
    p.log_n = log_n;
    p.t1 = (log_n + 1) / 2;
    p.t2 = log_n - p.t1;
    return p;
}
// table block of one (log n, direction): offsets in ELEMENTS (10 i32 each)
//   [0, 4)                      consts: w (the root of this direction), 1 / n (1 for the forward direction), 5, 1 / 5
//   LOC1 [N1 / 2]               w_N1^i          LOC2 [N2 / 2]  w_N2^i
//   TW_HI, TW_LO [2048 each]    w^(i << 11), w^i
//   IN_HI, IN_LO [2048 each]    5^(i << 11), 5^i                      (forward coset transform: x[j] * 5^j on the way in)
//   OUT_HI, OUT_LO [2048 each]  5^-(i << 11), 5^-i / n * 2^256        (OUT_LO[0] alone when there is no coset on the way out)
FRN_HOST_DEV u32 frn_off_loc1(const frn_plan &) { return 4; }
FRN_HOST_DEV u32 frn_off_loc2(const frn_plan &p) { return 4 + (1u << p.t1) / 2; }
FRN_HOST_DEV u32 frn_off_tw_hi(const frn_plan &p) { return frn_off_loc2(p) + ((1u << p.t2) / 2 ? (1u << p.t2) / 2 : 1); }
FRN_HOST_DEV u32 frn_off_tw_lo(const frn_plan &p) { return frn_off_tw_hi(p) + FRN_SPLIT_N; }
FRN_HOST_DEV u32 frn_off_in_hi(const frn_plan &p) { return frn_off_tw_lo(p) + FRN_SPLIT_N; }
FRN_HOST_DEV u32 frn_off_in_lo(const frn_plan &p) { return frn_off_in_hi(p) + FRN_SPLIT_N; }
FRN_HOST_DEV u32 frn_off_out_hi(const frn_plan &p) { return frn_off_in_lo(p) + FRN_SPLIT_N; }
FRN_HOST_DEV u32 frn_off_out_lo(const frn_plan &p) { return frn_off_out_hi(p) + FRN_SPLIT_N; }
FRN_HOST_DEV u32 frn_table_elems(const frn_plan &p) { return frn_off_out_lo(p) + FRN_SPLIT_N; }

ZKLC_HD fr frn_pow(fr a, u64 e) {
    const fr one = FR_ONE;
    fr r = one;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    while (e) {
        if (e & 1) r = fr_mul(r, a);
        a = fr_sqr(a);
        e >>= 1;
    }
    return r;
}
ZKLC_HD void frn_store(i32 *dst, const fr &a) {
#pragma unroll
    for (int k = 0; k < 10; k++) dst[k] = a.v[k];
}
ZKLC_HD fr frn_load(const i32 *src) {
    fr a;
#pragma unroll
    for (int k = 0; k < 10; k++) a.v[k] = src[k];
    return a;
}
ZKLC_HD u32 frn_bitrev(u32 i, u32 bits) {
    u32 r = 0;
    for (u32 b = 0; b < bits; b++) r |= ((i >> b) & 1u) << (bits - 1 - b);
    return r;
}

// the four constants of a table block (three inversions: run by ONE lane, once per (context, log n, direction))
ZKLC_HD void frn_table_consts(i32 *tab, u32 log_n, u32 inverse) {
    const u32 rw[8] = FR_ROOT28_WORDS;
    const fr r2 = FR_R2, one = FR_ONE;
    fr w = frn_pow(fr_from_regular(rw), 1ULL << (28 - log_n));
    fr nn = fr_zero();
    nn.v[0] = (i32)((1u << log_n) & 0x3ffffff);
    nn.v[1] = (i32)((1u << log_n) >> 26);
    fr five = fr_zero();
    five.v[0] = 5;
    five = fr_mul(five, r2);
    frn_store(tab, inverse ? fr_inv(w) : w);
    frn_store(tab + 10, inverse ? fr_inv(fr_mul(nn, r2)) : one);
    frn_store(tab + 20, five);
    frn_store(tab + 30, fr_inv(five));
}
// entry e (>= 4) of the table block, from its constants
ZKLC_HD fr frn_table_entry(const i32 *tab, const frn_plan &p, u32 e) {
    const fr c2p256 = FR_2P256;
    fr w = frn_load(tab), ninv = frn_load(tab + 10), g = frn_load(tab + 20), gi = frn_load(tab + 30);
    u32 i;
    if (e < frn_off_loc2(p)) return frn_pow(w, (u64)(e - frn_off_loc1(p)) << p.t2);            // w_N1 = w^N2
    if (e < frn_off_tw_hi(p)) return frn_pow(w, (u64)(e - frn_off_loc2(p)) << p.t1);           // w_N2 = w^N1
    if (e < frn_off_tw_lo(p)) return frn_pow(w, (u64)(e - frn_off_tw_hi(p)) << FRN_SPLIT);
    if (e < frn_off_in_hi(p)) return frn_pow(w, e - frn_off_tw_lo(p));
    if (e < frn_off_in_lo(p)) return frn_pow(g, (u64)(e - frn_off_in_hi(p)) << FRN_SPLIT);
    if (e < frn_off_out_hi(p)) return frn_pow(g, e - frn_off_in_lo(p));
    if (e < frn_off_out_lo(p)) return frn_pow(gi, (u64)(e - frn_off_out_hi(p)) << FRN_SPLIT);
    i = e - frn_off_out_lo(p);
    return fr_mul(fr_mul(frn_pow(gi, i), ninv), c2p256);
}

// ---- tiles: Nt = 2^T elements in limb-major order (limb k of element i at tile[k * Nt + i]: consecutive lanes, consecutive words)
ZKLC_HD fr frn_tile_load(const i32 *tile, u32 Nt, u32 i) {
    fr a;
#pragma unroll
    for (int k = 0; k < 10; k++) a.v[k] = tile[k * Nt + i];
    return a;
}
ZKLC_HD void frn_tile_store(i32 *tile, u32 Nt, u32 i, const fr &a) {
#pragma unroll
    for (int k = 0; k < 10; k++) tile[k * Nt + i] = a.v[k];
}
// butterfly b (< Nt / 2) of stage t (decimation in time on a bit-reversed tile): (u, v) <- (u + w v, u - w v), w = loc[j << (T - 1 - t)]
// with loc[i] = (tile root)^i.  Additions are lazy; stage 0 has w = 1 for every pair.
ZKLC_HD void frn_tile_butterfly(i32 *tile, u32 T, u32 t, u32 b, const i32 *loc) {
    const u32 Nt = 1u << T, half = 1u << t;
    u32 j = b & (half - 1);
    u32 i0 = ((b >> t) << (t + 1)) | j, i1 = i0 + half;
    fr u = frn_tile_load(tile, Nt, i0), v = frn_tile_load(tile, Nt, i1);
    if (t) v = fr_mul(v, frn_load(loc + (size_t)(j << (T - 1 - t)) * 10));
    frn_tile_store(tile, Nt, i0, fr_add(u, v));
    frn_tile_store(tile, Nt, i1, fr_sub(u, v));
}

ZKLC_HD fr frn_two_level(const i32 *hi, const i32 *lo, u32 e) {
    return fr_mul(frn_load(hi + (size_t)(e >> FRN_SPLIT) * 10), frn_load(lo + (size_t)(e & (FRN_SPLIT_N - 1)) * 10));
}
// pass A, element j1 of column j2: gnark words -> reduced internal value (times 5^j for the forward coset transform)
ZKLC_HD fr frn_pass_a_in(const u64 *data, const i32 *tab, const frn_plan &p, u32 j1, u32 j2, u32 coset_in) {
    const fr one = FR_ONE;
    u32 j = (j1 << p.t2) | j2;
    const u64 *q = data + (size_t)j * 4;
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        w[2 * k] = (u32)q[k];
        w[2 * k + 1] = (u32)(q[k] >> 32);
    }
    fr c = coset_in ? frn_two_level(tab + (size_t)frn_off_in_hi(p) * 10, tab + (size_t)frn_off_in_lo(p) * 10, j) : one;
    return fr_mul(fr_from_gnark(w), c);
}
// pass A, output k1 of column j2: times w^(j2 k1)
ZKLC_HD fr frn_pass_a_out(const fr &a, const i32 *tab, const frn_plan &p, u32 k1, u32 j2) {
    return fr_mul(a, frn_two_level(tab + (size_t)frn_off_tw_hi(p) * 10, tab + (size_t)frn_off_tw_lo(p) * 10, j2 * k1));
}
// pass B, output k2 of column k1 -> gnark words of X[k1 + N1 k2] (times 1 / n, and 5^-k after an inverse coset transform)
ZKLC_HD void frn_pass_b_out(u64 *data, const fr &a, const i32 *tab, const frn_plan &p, u32 k1, u32 k2, u32 coset_out) {
    u32 k = k1 | (k2 << p.t1);
    fr f = coset_out ? frn_two_level(tab + (size_t)frn_off_out_hi(p) * 10, tab + (size_t)frn_off_out_lo(p) * 10, k)
                     : frn_load(tab + (size_t)frn_off_out_lo(p) * 10);
    u32 w[8];
    fr_freeze_words(w, fr_mul(a, f));
    u64 *o = data + (size_t)k * 4;
#pragma unroll
    for (int q = 0; q < 4; q++) o[q] = (u64)w[2 * q] | ((u64)w[2 * q + 1] << 32);
}
