// GF(2^255-19) on 8 saturated 32-bit limbs (one field element = 8 VGPRs).
//
// Replaces the BigUint-per-operation field of the reference
// (crypto/plonky2_ed25519/src/field/ed25519_base.rs:19,99-116,178-230) for the
// native pre-verification at near_bft_finality/src/prove_block_data/signatures.rs:79.
//
// Representation: any value in [0, 2^256) congruent to the element mod p
// ("weakly reduced"); 2^256 = 38 (mod p) folds every overflow.  fe_freeze
// yields the unique canonical representative.  gfx950 has no 64x64 multiply:
// the schoolbook product runs on v_mad_u64_u32 (32x32+64) column sums.
#pragma once
#include "common.cuh"

// fe_mul / fe_sqr are real function calls on the device by default: a
// verification issues ~3200 of them and inlining every site produced ~290 KiB of
// ISA (far beyond the instruction cache).  Arguments travel by value in VGPRs.
#if defined(__HIPCC__) && !defined(ZKLC_FE_INLINE)
#define ZKLC_FE_CALL static __device__ __attribute__((noinline))
#else
#define ZKLC_FE_CALL ZKLC_HD
#endif

struct fe {
    u32 v[8];
};

ZKLC_HD fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
ZKLC_HD fe fe_one() {
    fe r = fe_zero();
    r.v[0] = 1;
    return r;
}

// r = a + b
ZKLC_HD fe fe_add(const fe &a, const fe &b) {
    fe r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a.v[i] + b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    // fold the carry (2^256 = 38); a second wrap is possible only from a tiny value
    c *= 38;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    r.v[0] += 38 * (u32)c;
    return r;
}

// r = a - b
ZKLC_HD fe fe_sub(const fe &a, const fe &b) {
    fe r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (u32)c;
        c >>= 32;  // arithmetic: 0 or -1
    }
    // borrow: subtract 38 (2^256 = 38)
    int64_t d = c * 38;  // 0 or -38
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d += r.v[i];
        r.v[i] = (u32)d;
        d >>= 32;
    }
    // a second borrow can only come from a value < 38: wraps to ~2^256, fix by -38 again
    r.v[0] -= 38 * (u32)(-d);
    return r;
}

ZKLC_HD fe fe_neg(const fe &a) { return fe_sub(fe_zero(), a); }

// fold a 512-bit product t[16] to 8 limbs
ZKLC_HD fe fe_fold(const u32 *t) {
    fe r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)t[i] + (u64)t[8 + i] * 38;
        r.v[i] = (u32)c;
        c >>= 32;
    }
    c *= 38;  // c <= 38 -> <= 1444
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    r.v[0] += 38 * (u32)c;  // if c==1 the value wrapped to < 1444: no further carry
    return r;
}

// Two formulations of the 8x8 limb product (selected per translation unit):
//   0: column (product-scanning) sums in a 96-bit accumulator
//   1: row (operand-scanning) sums; a*b + t + carry never overflows 64 bits,
//      so there is no carry flag at all -- gfx950 needs wait states between a
//      VALU write of VCC and its use as carry-in, which makes flag chains slow.
#ifndef ZKLC_FE_MUL_IMPL
#define ZKLC_FE_MUL_IMPL 1
#endif

ZKLC_FE_CALL fe fe_mul(const fe a, const fe b) {
    u32 t[16];
#if ZKLC_FE_MUL_IMPL == 0
    u64 lo = 0;
    u32 hi = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int j = k - i;
            if (j >= 0 && j < 8) mac96(lo, hi, a.v[i], b.v[j]);
        }
        t[k] = (u32)lo;
        lo = (lo >> 32) | ((u64)hi << 32);
        hi = 0;
    }
    t[15] = (u32)lo;
#else
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u64 x = (u64)a.v[i] * b.v[j] + (i ? t[i + j] : 0u) + carry;
            t[i + j] = (u32)x;
            carry = (u32)(x >> 32);
        }
        t[i + 8] = carry;
    }
#endif
    return fe_fold(t);
}

ZKLC_FE_CALL fe fe_sqr(const fe a) {
    // off-diagonal sums, doubled, plus the squares
    u32 t[16];
#if ZKLC_FE_MUL_IMPL == 0
    u64 lo = 0;
    u32 hi = 0;
    t[0] = 0;
#pragma unroll
    for (int k = 1; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int j = k - i;
            if (j > i && j < 8) mac96(lo, hi, a.v[i], a.v[j]);
        }
        t[k] = (u32)lo;
        lo = (lo >> 32) | ((u64)hi << 32);
        hi = 0;
    }
    t[15] = (u32)lo;
#else
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        u32 carry = 0;
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            u64 x = (u64)a.v[i] * a.v[j] + t[i + j] + carry;
            t[i + j] = (u32)x;
            carry = (u32)(x >> 32);
        }
        t[i + 8] = carry;
    }
#endif
    // t = 2*t + sum a_i^2 * 2^(64 i)
    u32 top = 0;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 sq = (u64)a.v[i] * a.v[i];
        u32 d0 = (t[2 * i] << 1) | top;
        top = t[2 * i] >> 31;
        u32 d1 = (t[2 * i + 1] << 1) | top;
        top = t[2 * i + 1] >> 31;
        c += (u64)d0 + (u32)sq;
        t[2 * i] = (u32)c;
        c >>= 32;
        c += (u64)d1 + (sq >> 32);
        t[2 * i + 1] = (u32)c;
        c >>= 32;
    }
    return fe_fold(t);
}

// r = a * small (small < 2^32)
ZKLC_HD fe fe_mul_small(const fe &a, u32 s) {
    fe r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a.v[i] * s;
        r.v[i] = (u32)c;
        c >>= 32;
    }
    c *= 38;  // c < 2^32 -> < 2^38
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    r.v[0] += 38 * (u32)c;
    return r;
}

// canonical representative in [0, p)
ZKLC_HD fe fe_freeze(const fe &a) {
    fe r = a;
    // fold bit 255: r = (r mod 2^255) + 19*(r >> 255)  ->  r < 2^255 + 19
    u64 c = (u64)(r.v[7] >> 31) * 19;
    r.v[7] &= 0x7fffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r.v[i];
        r.v[i] = (u32)c;
        c >>= 32;
    }
    // t = r + 19; if t >= 2^255 then r >= p and r - p = t - 2^255
    fe t;
    c = 19;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r.v[i];
        t.v[i] = (u32)c;
        c >>= 32;
    }
    u32 ge = t.v[7] >> 31;  // 1 if r >= p
    t.v[7] &= 0x7fffffffu;
    u32 m = 0u - ge;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (t.v[i] & m) | (r.v[i] & ~m);
    return r;
}

// 1 if a == 0 (mod p)
ZKLC_HD u32 fe_is_zero(const fe &a) {
    fe f = fe_freeze(a);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= f.v[i];
    return o == 0;
}

ZKLC_HD u32 fe_eq(const fe &a, const fe &b) { return fe_is_zero(fe_sub(a, b)); }

// parity of the canonical representative
ZKLC_HD u32 fe_is_negative(const fe &a) { return fe_freeze(a).v[0] & 1; }

// r = cond ? b : a   (cond in {0,1})
ZKLC_HD fe fe_select(const fe &a, const fe &b, u32 cond) {
    fe r;
    u32 m = 0u - cond;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
}

// little-endian 32 bytes -> element; the top bit is masked off (the caller
// keeps it as the sign of x).  Values >= p are accepted un-reduced, exactly
// as curve25519-dalek's FieldElement::from_bytes does.
ZKLC_HD fe fe_from_words(const u32 *w) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = w[i];
    r.v[7] &= 0x7fffffffu;
    return r;
}

ZKLC_HD fe fe_sqr_n(fe a, int n) {
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int i = 0; i < n; i++) a = fe_sqr(a);
    return a;
}

// a^(2^250 - 1) and a^11, the shared prefix of inversion / sqrt chains
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

// edwards25519 constants, little-endian 32-bit limbs
// d  = 37095705934669439343138083508754565189542113879843219016388785533085940283555
//      (crypto/plonky2_ed25519/src/curve/ed25519.rs:24-29)
#define FE_D {{0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu}}
#define FE_2D {{0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu}}
// sqrt(-1) = 2^((p-1)/4)
#define FE_SQRTM1 {{0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u}}
