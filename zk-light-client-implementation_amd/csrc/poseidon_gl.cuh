// Poseidon permutation over Goldilocks: width 12, rate 8, x^7, 4 + 22 + 4 rounds.
//
// Replaces plonky2::hash::poseidon (plonky2-near@2244a9d, un-vendored) as restated by
// gnark-plonky2-verifier/poseidon/goldilocks.go:30-37 (Poseidon), :92-115 (rounds),
// :138-145 (S-box), :172-216 (MDS = circulant [17,15,41,16,2,28,13,13,39,18,34,20]
// + diag [8,0,..]), :231-331 (fast partial rounds); sponge :41-86.
// One lane = one state (24 VGPRs).  The MDS entries are 6-bit, so a row is 13
// 32x6-bit multiply-adds on the low halves and 13 on the high halves (no reduction
// until the end of the row); the round constants sit in constant memory and are
// read with wave-uniform (scalar) loads.
#pragma once
#include "goldilocks.cuh"

#if defined(__HIPCC__)
#define ZKLC_CONST_ARRAY __device__ __constant__ const
#else
#define ZKLC_CONST_ARRAY static const
#endif
#include "poseidon_gl_constants.inc"

ZKLC_HD u64 pgl_sbox(u64 x) {
    u64 x2 = gl_sqr(x);
    u64 x3 = gl_mul(x2, x);
    u64 x4 = gl_sqr(x2);
    return gl_mul(x3, x4);
}

ZKLC_HD void pgl_mds(u64 *s) {
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (u32)s[i];
        hi[i] = (u32)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 sl = 0, sh = 0;  // each < 2^32 * 264
#pragma unroll
        for (int i = 0; i < 12; i++) {
            sl += (u64)lo[(i + r) % 12] * C[i];
            sh += (u64)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) {
            sl += (u64)lo[0] * 8;
            sh += (u64)hi[0] * 8;
        }
        // value = sl + sh * 2^32 as (hi:lo)
        u64 l = sl + (sh << 32);
        u64 h = (sh >> 32) + (l < sl);
        s[r] = gl_reduce128(l, h);
    }
}

ZKLC_HD void pgl_full_round(u64 *s, int rnd) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = pgl_sbox(gl_add(s[i], PGL_RC[12 * rnd + i]));
    pgl_mds(s);
}

// ---- the permutation proper works on LOOSE values (any u64 congruent to the element; goldilocks.cuh) and makes its
// output canonical at the end: 5 instructions fewer per multiplication, and the sums of products of the partial
// rounds are accumulated in 160 bits and reduced once (plonky2 does the same on the CPU with u128 sums).
ZKLC_HD u64 pgl_sbox_l(u64 x) {
    u64 x2 = gl_mul_loose(x, x);
    u64 x3 = gl_mul_loose(x2, x);
    u64 x4 = gl_mul_loose(x2, x2);
    return gl_mul_loose(x3, x4);
}
ZKLC_HD void pgl_mds_l(u64 *s) {
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u32 lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (u32)s[i];
        hi[i] = (u32)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u64 sl = 0, sh = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            sl += (u64)lo[(i + r) % 12] * C[i];
            sh += (u64)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) {
            sl += (u64)lo[0] * 8;
            sh += (u64)hi[0] * 8;
        }
        u64 l = sl + (sh << 32);
        u64 h = (sh >> 32) + (l < sl);
        s[r] = gl_reduce128_loose(l, h);
    }
}
ZKLC_HD void pgl_full_round_l(u64 *s, int rnd) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = pgl_sbox_l(gl_add_lc(s[i], PGL_RC[12 * rnd + i]));
    pgl_mds_l(s);
}

ZKLC_HD void poseidon_gl_permute(u64 *s) {
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) pgl_full_round_l(s, r);
    // partial rounds in the "fast" form (goldilocks.go:102-115, :231-331)
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_lc(s[i], PGL_FP_FIRST[i]);
    {
        // t[d] = sum_{r=1..11} s[r] * INIT[r-1][d-1]; one output per iteration, rotated into place (static register indices)
        u64 t[12];
#pragma unroll
        for (int d = 1; d < 12; d++) t[d] = 0;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (int d = 1; d < 12; d++) {
            gl_acc160 acc = {0, 0, 0};
#pragma unroll
            for (int r = 1; r < 12; r++) gl_acc_mul(acc, s[r], PGL_FP_INIT[(r - 1) * 11 + d - 1]);
#pragma unroll
            for (int q = 1; q < 11; q++) t[q] = t[q + 1];
            t[11] = gl_acc_reduce(acc);
        }
#pragma unroll
        for (int i = 1; i < 12; i++) s[i] = t[i];
    }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int i = 0; i < 22; i++) {
        u64 s0 = gl_add_lc(pgl_sbox_l(s[0]), PGL_FP_RC[i]);
        gl_acc160 acc = {0, 0, 0};
        gl_acc_mul(acc, s0, 25);  // MDS0TO0
#pragma unroll
        for (int j = 1; j < 12; j++) gl_acc_mul(acc, s[j], PGL_FP_WHATS[i * 11 + j - 1]);
#pragma unroll
        for (int j = 1; j < 12; j++) {
            // s[j] + s0 * v < 2^128: one reduction of the sum
            u64 lo, hi;
            gl_mul_wide(s0, PGL_FP_VS[i * 11 + j - 1], lo, hi);
            u64 l2 = lo + s[j];
            hi += (l2 < lo);
            s[j] = gl_reduce128_loose(l2, hi);
        }
        s[0] = gl_acc_reduce(acc);
    }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) pgl_full_round_l(s, 26 + r);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canonical(s[i]);
}

// hash_or_noop / hash_no_pad of `len` elements read through a strided accessor:
// element i is in[i * stride].  Overwrite-mode sponge, rate 8, no padding
// (goldilocks.go:41-86); inputs of <= 4 elements are returned padded (hash_or_noop).
ZKLC_HD void poseidon_gl_hash_or_noop(const u64 *in, size_t stride, u32 len, u64 *out4) {
    if (len <= 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) out4[i] = (u32)i < len ? in[(size_t)i * stride] : 0;
        return;
    }
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (u32 off = 0; off < len; off += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (off + j < len) s[j] = in[(size_t)(off + j) * stride];
        poseidon_gl_permute(s);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out4[i] = s[i];
}

// two_to_one(l, r) = permute(l || r || 0000)[0..4]
ZKLC_HD void poseidon_gl_two_to_one(const u64 *l, const u64 *r, u64 *out4) {
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        s[i] = l[i];
        s[4 + i] = r[i];
        s[8 + i] = 0;
    }
    poseidon_gl_permute(s);
#pragma unroll
    for (int i = 0; i < 4; i++) out4[i] = s[i];
}
