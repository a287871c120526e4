// Goldilocks field p = 2^64 - 2^32 + 1 (canonical u64 in, canonical u64 out).
//
// Replaces plonky2_field::goldilocks_field (plonky2-near@2244a9d, un-vendored:
// Cargo.toml:44-47); parameters as restated in
// gnark-plonky2-verifier/goldilocks/base.go:33-42 (generator 7, 2-adicity 32,
// POWER_OF_TWO_GENERATOR 1753635133440165772).  gfx950 has no 64x64 multiply:
// a product is four v_mad_u64_u32 and the reduction uses 2^64 = 2^32 - 1,
// 2^96 = -1 (mod p).
#pragma once
#include "common.cuh"

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
#define GL_GENERATOR 7ULL
#define GL_POWER_OF_TWO_GENERATOR 1753635133440165772ULL

// The three primitives below are written so that hipcc emits the short forms (gfx950: v_lshl_add_u64 is a one-instruction 64-bit
// add, v_cmp_*_u64 a one-instruction compare): "add 2^32 - 1 if the sum wrapped OR is >= p" is ONE select + ONE add for both
// conditions (wrapped: 2^64 = EPS; >= p: s - p = s + EPS mod 2^64).  5 / 6 / ~10 instructions; the textbook forms with two selects
// were 9 / 6 / 14 (tools/ubench counts; every prover kernel is integer-issue-bound).
ZKLC_HD u64 gl_add(u64 a, u64 b) {
    u64 s = a + b;
    // canonical inputs: a wrapped sum s + 2^64 = s + EPS cannot wrap again and is < p; an unwrapped s >= p is < 2p
    bool fix = (s < a) | (s >= GL_P);
    return s + (fix ? GL_EPS : 0);
}
ZKLC_HD u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b;
    return d - ((a < b) ? GL_EPS : 0);      // borrow: d + p = d - EPS (mod 2^64)
}
ZKLC_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
ZKLC_HD u64 gl_double(u64 a) { return gl_add(a, a); }

// (hi:lo) mod p for any 128-bit value
ZKLC_HD u64 gl_reduce128(u64 lo, u64 hi) {
    u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    u64 t0 = lo - hi_hi;                      // 2^96 = -1
    t0 -= (lo < hi_hi) ? GL_EPS : 0;          // borrow: -2^64 = -EPS (no second borrow: lo - hi_hi + 2^64 > EPS)
    u64 r = t0 + hi_lo * GL_EPS;              // 2^64 = EPS ; the product is <= 2^64 - 2^33 + 1 (one v_mad_u64_u32 with t0 as addend)
    // carry: r + EPS <= p - 2, no second carry; no carry and r >= p: r - p = r + EPS (mod 2^64); never both (a carried r is < p)
    bool fix = (r < t0) | (r >= GL_P);
    return r + (fix ? GL_EPS : 0);
}

// "Loose" arithmetic: values are any u64 congruent to the field element (not necessarily < p).  Products and reduce128
// accept loose inputs; only additions need care.  Used inside the Poseidon permutation, canonicalised on the way out.
ZKLC_HD u64 gl_reduce128_loose(u64 lo, u64 hi) {
    u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS;
    u64 t1 = hi_lo * GL_EPS;
    u64 r = t0 + t1;
    if (r < t1) r += GL_EPS;  // cannot carry again (see gl_reduce128)
    return r;
}
// loose + canonical -> loose (a single wrap is possible: a < 2^64, b < p => a + b - 2^64 + EPS < 2^64)
ZKLC_HD u64 gl_add_lc(u64 a_loose, u64 b_canonical) {
    u64 s = a_loose + b_canonical;
    return s + ((s < a_loose) ? GL_EPS : 0);
}
ZKLC_HD u64 gl_canonical(u64 a) { return a >= GL_P ? a - GL_P : a; }

ZKLC_HD void gl_mul_wide(u64 a, u64 b, u64 &lo, u64 &hi) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);           // < 2^64: (2^32-1)^2 + 2^32 - 1
    u64 p10 = (u64)a1 * b0 + (u32)p01;              // < 2^64
    u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    lo = (p10 << 32) | (u32)p00;
    hi = p11;
}

ZKLC_HD u64 gl_mul(u64 a, u64 b) {
    u64 lo, hi;
    gl_mul_wide(a, b, lo, hi);
    return gl_reduce128(lo, hi);
}
ZKLC_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }
ZKLC_HD u64 gl_mul_loose(u64 a, u64 b) {
    u64 lo, hi;
    gl_mul_wide(a, b, lo, hi);
    return gl_reduce128_loose(lo, hi);
}
// 160-bit accumulator for sums of 128-bit products (lo, hi, number of 2^128 overflows)
struct gl_acc160 {
    u64 lo, hi;
    u32 over;
};
ZKLC_HD void gl_acc_mul(gl_acc160 &a, u64 x, u64 y) {
    u64 plo, phi;
    gl_mul_wide(x, y, plo, phi);
    u64 lo = a.lo + plo;
    u64 c = lo < plo;
    u64 hi = a.hi + phi;
    u32 o = hi < phi;
    hi += c;
    o += hi < c;
    a.lo = lo;
    a.hi = hi;
    a.over += o;
}
// 2^128 = 2^96 * 2^32 = -2^32 (mod p)
ZKLC_HD u64 gl_acc_reduce(const gl_acc160 &a) {
    u64 r = gl_reduce128(a.lo, a.hi);
    return gl_sub(r, (u64)a.over << 32);
}

// Carry-free accumulation of x * k for many (x, k) where k comes from a TABLE: the table holds the 22-bit limbs of k and of
// k' = 2^32 k mod p ({ka, kb, kc, k'a, k'b, k'c}, gl_limbs22), so that  x k = x0 k + x1 k'  (mod p) is six 32 x 22-bit products
// added into three 64-bit columns of weights 1, 2^22, 2^44 -- six v_mad_u64_u32 and no carry handling at all (a 128-bit
// product accumulated with carries costs ~20 instructions).  A column holds 2^10 products: callers fold before 500 terms.
struct gl_acc3 {
    u64 c0, c1, c2;
};
// A uniform read-only table in global memory, addressed through the CONSTANT address space: the compiler may then fetch it with
// scalar loads (s_load_dwordx8/x16 into SGPRs, batched over consecutive entries).  Through a plain pointer the same reads are
// per-lane global loads whose latency sits in front of every use, because a kernel that also stores cannot prove them invariant.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(4))) const u32 gl_ktab;
typedef __attribute__((address_space(4))) const u64 gl_ktab64;
#else
typedef const u32 gl_ktab;
typedef const u64 gl_ktab64;
#endif
ZKLC_HD void gl_limbs22(u64 k, u32 *out6) {
    u64 k2 = gl_mul(k, 1ULL << 32);
    out6[0] = (u32)(k & 0x3FFFFF);
    out6[1] = (u32)((k >> 22) & 0x3FFFFF);
    out6[2] = (u32)(k >> 44);
    out6[3] = (u32)(k2 & 0x3FFFFF);
    out6[4] = (u32)((k2 >> 22) & 0x3FFFFF);
    out6[5] = (u32)(k2 >> 44);
}
template <class K>
ZKLC_HD void gl_acc3_mul(gl_acc3 &a, u64 x, K k6) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    a.c0 += (u64)x0 * k6[0];
    a.c1 += (u64)x0 * k6[1];
    a.c2 += (u64)x0 * k6[2];
    a.c0 += (u64)x1 * k6[3];
    a.c1 += (u64)x1 * k6[4];
    a.c2 += (u64)x1 * k6[5];
}
// c0 + c1 2^22 + c2 2^44 mod p, canonical (any 64-bit columns)
ZKLC_HD u64 gl_acc3_reduce(const gl_acc3 &a) {
    u64 t1 = a.c1 << 22, t2 = a.c2 << 44;
    u64 lo = a.c0 + t1;
    u64 hi = (a.c1 >> 42) + (a.c2 >> 20) + (lo < t1);
    lo += t2;
    hi += lo < t2;
    return gl_reduce128(lo, hi);
}
// the accumulator restarted from its own value (columns back below 2^22)
ZKLC_HD void gl_acc3_normalize(gl_acc3 &a) {
    u64 v = gl_acc3_reduce(a);
    a.c0 = v & 0x3FFFFF;
    a.c1 = (v >> 22) & 0x3FFFFF;
    a.c2 = v >> 44;
}

ZKLC_HD u64 gl_pow(u64 a, u64 e) {
    u64 r = 1;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_sqr(a);
        e >>= 1;
    }
    return r;
}
ZKLC_HD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
// primitive 2^log_n-th root of unity
ZKLC_HD u64 gl_root_of_unity(u32 log_n) { return gl_pow(GL_POWER_OF_TWO_GENERATOR, 1ULL << (32 - log_n)); }
