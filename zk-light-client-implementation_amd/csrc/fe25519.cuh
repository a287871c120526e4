// GF(2^255-19) in radix 2^25.5: ten signed 32-bit limbs of alternating 26/25
// bits (value = sum h[i] * 2^ceil(25.5 i)), 64-bit signed column sums.
//
// Replaces the BigUint-per-operation field of the reference
// (crypto/plonky2_ed25519/src/field/ed25519_base.rs:19,99-116,178-230) for the
// native pre-verification at near_bft_finality/src/prove_block_data/signatures.rs:79.
//
// Why this radix on gfx950 (profiles/r01_valu_ubench.txt): v_mad_u64_u32 /
// v_mad_i64_i32 issue at the same rate as v_add3_u32, but a VALU carry flag
// (VCC) needs wait states before it can be consumed, so saturated 32-bit limbs
// pay for every carry.  With 25.5-bit limbs a whole product column (10 terms,
// the 2^255 = 19 wrap folded in) accumulates in ONE 64-bit register by a chain
// of v_mad_i64_i32 -- no carries until the single propagation pass at the end.
//
// Bounds (the classic ref10 analysis):
//   "reduced"  = output of mul/sqr/sqr2/carry: |h| <= 1.01*2^25 (even i), 1.01*2^24 (odd i)
//   mul/sqr inputs may be sums/differences of up to THREE reduced values
//   (|f| <= 1.65*2^26 / 1.65*2^25); add/sub/neg do no carrying.
#pragma once
#include "common.cuh"

// fe_mul / fe_sqr / fe_sqr2 are real function calls on the device: a verification
// issues ~3000 of them and inlining every site produced ~290 KiB of ISA (far
// beyond the instruction cache).  Arguments travel by value in VGPRs.
#if defined(__HIPCC__) && !defined(ZKLC_FE_INLINE)
#define ZKLC_FE_CALL static __device__ __attribute__((noinline))
#else
#define ZKLC_FE_CALL ZKLC_HD
#endif

typedef int32_t i32;
typedef int64_t i64;

struct fe {
    i32 v[10];
};

#if defined(ZKLC_FE_BOUND_CHECKS)
#include <assert.h>
static inline void fe_check_in(const fe &a) {  // precondition of mul/sqr
    for (int i = 0; i < 10; i++) {
        i64 lim = (i & 1) ? (i64)(1.65 * (1 << 25)) : (i64)(1.65 * (1 << 26));
        assert(a.v[i] <= lim && a.v[i] >= -lim);
    }
}
#else
#define fe_check_in(a) ((void)0)
#endif

ZKLC_HD fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = 0;
    return r;
}
ZKLC_HD fe fe_one() {
    fe r = fe_zero();
    r.v[0] = 1;
    return r;
}
ZKLC_HD fe fe_add(const fe &a, const fe &b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
ZKLC_HD fe fe_sub(const fe &a, const fe &b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] - b.v[i];
    return r;
}
ZKLC_HD fe fe_neg(const fe &a) {
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = -a.v[i];
    return r;
}

// carry propagation of ten 64-bit column sums into a reduced element
// (ref10 order: two interleaved chains 0->1->2->3->4 and 4->5->...->9->0)
ZKLC_HD fe fe_carry64(i64 *h) {
#define ZKLC_CARRY_EVEN(i, j)                     \
    {                                             \
        i64 c = (h[i] + ((i64)1 << 25)) >> 26;    \
        h[j] += c;                                \
        h[i] -= c << 26;                          \
    }
#define ZKLC_CARRY_ODD(i, j)                      \
    {                                             \
        i64 c = (h[i] + ((i64)1 << 24)) >> 25;    \
        h[j] += c;                                \
        h[i] -= c << 25;                          \
    }
    ZKLC_CARRY_EVEN(0, 1)
    ZKLC_CARRY_EVEN(4, 5)
    ZKLC_CARRY_ODD(1, 2)
    ZKLC_CARRY_ODD(5, 6)
    ZKLC_CARRY_EVEN(2, 3)
    ZKLC_CARRY_EVEN(6, 7)
    ZKLC_CARRY_ODD(3, 4)
    ZKLC_CARRY_ODD(7, 8)
    ZKLC_CARRY_EVEN(4, 5)
    ZKLC_CARRY_EVEN(8, 9)
    {
        i64 c = (h[9] + ((i64)1 << 24)) >> 25;
        h[0] += c * 19;
        h[9] -= c << 25;
    }
    ZKLC_CARRY_EVEN(0, 1)
#undef ZKLC_CARRY_EVEN
#undef ZKLC_CARRY_ODD
    fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = (i32)h[i];
    return r;
}

// f_i * g_j lands in column (i+j) mod 10, times 19 when it wraps (2^255 = 19)
// and times 2 when both limbs are odd (their weights are 2^(25.5 i) rounded up).
ZKLC_FE_CALL fe fe_mul(const fe f, const fe g) {
    fe_check_in(f);
    fe_check_in(g);
    i32 g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        g19[i] = 19 * g.v[i];
        f2[i] = 2 * f.v[i];
    }
    i64 h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        i64 acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            int j = (k - i + 10) % 10;
            bool wrap = (i + j) >= 10;
            bool both_odd = (i & 1) && (j & 1);
            acc += (i64)(both_odd ? f2[i] : f.v[i]) * (wrap ? g19[j] : g.v[j]);
        }
        h[k] = acc;
    }
    return fe_carry64(h);
}

template <bool TWICE>
ZKLC_HD fe fe_sqr_impl(const fe &f) {
    fe_check_in(f);
    i32 f2[10], f19[10], f38[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        f2[i] = 2 * f.v[i];
        f19[i] = 19 * f.v[i];
        f38[i] = 38 * f.v[i];
    }
    i64 h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        i64 acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            int j = (k - i + 10) % 10;
            if (j < i) continue;
            bool wrap = (i + j) >= 10;
            bool both_odd = (i & 1) && (j & 1);
            if (i == j) {
                // f_i^2 * (2 if odd) * (19 if wrap)
                i32 b = wrap ? (both_odd ? f38[i] : f19[i]) : (both_odd ? f2[i] : f.v[i]);
                acc += (i64)f.v[i] * b;
            } else {
                // 2 f_i f_j * (2 if both odd) * (19 if wrap)
                i32 b = wrap ? (both_odd ? f38[j] : f19[j]) : (both_odd ? f2[j] : f.v[j]);
                acc += (i64)f2[i] * b;
            }
        }
        h[k] = TWICE ? acc + acc : acc;
    }
    return fe_carry64(h);
}
ZKLC_FE_CALL fe fe_sqr(const fe f) { return fe_sqr_impl<false>(f); }
// 2 * f^2, reduced (keeps the doubling formula inside the three-term bound)
ZKLC_FE_CALL fe fe_sqr2(const fe f) { return fe_sqr_impl<true>(f); }

// canonical value as 8 little-endian 32-bit words (ref10 fe_tobytes)
ZKLC_HD void fe_freeze_words(u32 *out, const fe &a) {
    i32 h[10];
#pragma unroll
    for (int i = 0; i < 10; i++) h[i] = a.v[i];
    i32 q = (19 * h[9] + ((i32)1 << 24)) >> 25;
#pragma unroll
    for (int i = 0; i < 10; i++) q = (h[i] + q) >> ((i & 1) ? 25 : 26);
    h[0] += 19 * q;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int w = (i & 1) ? 25 : 26;
        i32 c = h[i] >> w;
        h[i + 1] += c;
        h[i] -= c << w;
    }
    h[9] &= (1 << 25) - 1;
    // bit offsets 0,26,51,77,102,128,153,179,204,230
    u32 u[10];
#pragma unroll
    for (int i = 0; i < 10; i++) u[i] = (u32)h[i];
    out[0] = u[0] | (u[1] << 26);
    out[1] = (u[1] >> 6) | (u[2] << 19);
    out[2] = (u[2] >> 13) | (u[3] << 13);
    out[3] = (u[3] >> 19) | (u[4] << 6);
    out[4] = u[5] | (u[6] << 25);
    out[5] = (u[6] >> 7) | (u[7] << 19);
    out[6] = (u[7] >> 13) | (u[8] << 12);
    out[7] = (u[8] >> 20) | (u[9] << 6);
}

// little-endian 8 words -> element; bit 255 is ignored (the caller keeps it as
// the sign of x).  Values >= p are accepted un-reduced, exactly as
// curve25519-dalek's FieldElement::from_bytes does.
ZKLC_HD fe fe_from_words(const u32 *w) {
    fe r;
    r.v[0] = (i32)(w[0] & 0x3ffffff);
    r.v[1] = (i32)(((w[0] >> 26) | (w[1] << 6)) & 0x1ffffff);
    r.v[2] = (i32)(((w[1] >> 19) | (w[2] << 13)) & 0x3ffffff);
    r.v[3] = (i32)(((w[2] >> 13) | (w[3] << 19)) & 0x1ffffff);
    r.v[4] = (i32)((w[3] >> 6) & 0x3ffffff);
    r.v[5] = (i32)(w[4] & 0x1ffffff);
    r.v[6] = (i32)(((w[4] >> 25) | (w[5] << 7)) & 0x3ffffff);
    r.v[7] = (i32)(((w[5] >> 19) | (w[6] << 13)) & 0x1ffffff);
    r.v[8] = (i32)(((w[6] >> 12) | (w[7] << 20)) & 0x3ffffff);
    r.v[9] = (i32)((w[7] >> 6) & 0x1ffffff);
    return r;
}

// canonical limbs (each in [0, 2^26) / [0, 2^25))
ZKLC_HD fe fe_freeze(const fe &a) {
    u32 w[8];
    fe_freeze_words(w, a);
    return fe_from_words(w);
}

ZKLC_HD u32 fe_is_zero(const fe &a) {
    u32 w[8];
    fe_freeze_words(w, a);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= w[i];
    return o == 0;
}
ZKLC_HD u32 fe_eq(const fe &a, const fe &b) { return fe_is_zero(fe_sub(a, b)); }
ZKLC_HD u32 fe_is_negative(const fe &a) {
    u32 w[8];
    fe_freeze_words(w, a);
    return w[0] & 1;
}

// r = cond ? b : a   (cond in {0,1})
ZKLC_HD fe fe_select(const fe &a, const fe &b, u32 cond) {
    fe r;
    i32 m = -(i32)cond;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
}

ZKLC_HD fe fe_sqr_n(fe a, int n) {
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int i = 0; i < n; i++) a = fe_sqr(a);
    return a;
}

// a^(2^250 - 1) and a^11, the shared prefix of the inversion / sqrt chains
ZKLC_HD void fe_pow_2_250_1(const fe &z, fe &z_250_0, fe &z11) {
    fe z2 = fe_sqr(z);
    fe z8 = fe_sqr_n(z2, 2);
    fe z9 = fe_mul(z, z8);
    z11 = fe_mul(z2, z9);
    fe z22 = fe_sqr(z11);
    fe z_5_0 = fe_mul(z9, z22);
    fe z_10_0 = fe_mul(fe_sqr_n(z_5_0, 5), z_5_0);
    fe z_20_0 = fe_mul(fe_sqr_n(z_10_0, 10), z_10_0);
    fe z_40_0 = fe_mul(fe_sqr_n(z_20_0, 20), z_20_0);
    fe z_50_0 = fe_mul(fe_sqr_n(z_40_0, 10), z_10_0);
    fe z_100_0 = fe_mul(fe_sqr_n(z_50_0, 50), z_50_0);
    fe z_200_0 = fe_mul(fe_sqr_n(z_100_0, 100), z_100_0);
    z_250_0 = fe_mul(fe_sqr_n(z_200_0, 50), z_50_0);
}

// a^(p-2)
ZKLC_HD fe fe_invert(const fe &z) {
    fe t, z11;
    fe_pow_2_250_1(z, t, z11);
    return fe_mul(fe_sqr_n(t, 5), z11);
}

// a^((p-5)/8) = a^(2^252 - 3)
ZKLC_HD fe fe_pow22523(const fe &z) {
    fe t, z11;
    fe_pow_2_250_1(z, t, z11);
    return fe_mul(fe_sqr_n(t, 2), z);
}

// edwards25519 constants in radix 2^25.5
// d = 37095705934669439343138083508754565189542113879843219016388785533085940283555
//     (crypto/plonky2_ed25519/src/curve/ed25519.rs:24-29)
#define FE_D {{56195235, 13857412, 51736253, 6949390, 114729, 24766616, 60832955, 30306712, 48412415, 21499315}}
#define FE_2D {{45281625, 27714825, 36363642, 13898781, 229458, 15978800, 54557047, 27058993, 29715967, 9444199}}
// sqrt(-1) = 2^((p-1)/4)
#define FE_SQRTM1 {{34513072, 25610706, 9377949, 3500415, 12389472, 33281959, 41962654, 31548777, 326685, 11406482}}
