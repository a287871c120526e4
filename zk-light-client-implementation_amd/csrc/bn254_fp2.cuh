// Fp2 = Fp[u] / (u^2 + 1) for BN254 G2 and the pairing tower, on the lazy ten-limb Montgomery Fp of bn254_fp.cuh.
// Replaces gnark-crypto's `fptower.E2` (ecc/bn254/internal/fptower, un-vendored: gnark-plonky2-verifier/go.mod:9);
// memory layout of an E2 at the ABI = A0 then A1, each 4 little-endian u64 in Montgomery form.
// A product accumulates the columns of both partial products before ONE Montgomery reduction per coordinate
// (a0 b0 - a1 b1 and a0 b1 + a1 b0: 400 multiply-adds + 2 reductions instead of 4 reductions).
#pragma once
#include "bn254_fp.cuh"

struct fp2 {
    fp c0, c1;
};

ZKLC_HD fp2 fp2_zero() {
    fp2 r;
    r.c0 = fp_zero();
    r.c1 = fp_zero();
    return r;
}
ZKLC_HD fp2 fp2_one() {
    const fp one = FP_ONE;
    fp2 r;
    r.c0 = one;
    r.c1 = fp_zero();
    return r;
}
ZKLC_HD fp2 fp2_add(const fp2 &a, const fp2 &b) {
    fp2 r;
    r.c0 = fp_add(a.c0, b.c0);
    r.c1 = fp_add(a.c1, b.c1);
    return r;
}
ZKLC_HD fp2 fp2_sub(const fp2 &a, const fp2 &b) {
    fp2 r;
    r.c0 = fp_sub(a.c0, b.c0);
    r.c1 = fp_sub(a.c1, b.c1);
    return r;
}
ZKLC_HD fp2 fp2_neg(const fp2 &a) {
    fp2 r;
    r.c0 = fp_neg(a.c0);
    r.c1 = fp_neg(a.c1);
    return r;
}
ZKLC_HD fp2 fp2_dbl(const fp2 &a) { return fp2_add(a, a); }
ZKLC_HD fp2 fp2_conj(const fp2 &a) {
    fp2 r;
    r.c0 = a.c0;
    r.c1 = fp_neg(a.c1);
    return r;
}
ZKLC_HD fp2 fp2_select(const fp2 &a, const fp2 &b, u32 cond) {
    fp2 r;
    r.c0 = fp_select(a.c0, b.c0, cond);
    r.c1 = fp_select(a.c1, b.c1, cond);
    return r;
}
ZKLC_HD fp2 fp2_reduce(const fp2 &a) {
    fp2 r;
    r.c0 = fp_reduce(a.c0);
    r.c1 = fp_reduce(a.c1);
    return r;
}
ZKLC_HD u32 fp2_is_zero(const fp2 &a) { return fp_is_zero(a.c0) & fp_is_zero(a.c1); }

ZKLC_HD fp2 fp2_mul(const fp2 &a, const fp2 &b) {
    i64 t0[20], t1[20];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        i64 s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            int j = k - i;
            if (j >= 0 && j < 10) {
                s0 += (i64)a.c0.v[i] * b.c0.v[j] - (i64)a.c1.v[i] * b.c1.v[j];
                s1 += (i64)a.c0.v[i] * b.c1.v[j] + (i64)a.c1.v[i] * b.c0.v[j];
            }
        }
        t0[k] = s0;
        t1[k] = s1;
    }
    t0[19] = t1[19] = 0;
    fp2 r;
    r.c0 = fp_mont_reduce(t0);
    r.c1 = fp_mont_reduce(t1);
    return r;
}
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u
ZKLC_HD fp2 fp2_sqr(const fp2 &a) {
    fp2 r;
    r.c0 = fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1));
    r.c1 = fp_mul(fp_dbl(a.c0), a.c1);
    return r;
}
ZKLC_HD fp2 fp2_mul_fp(const fp2 &a, const fp &s) {
    fp2 r;
    r.c0 = fp_mul(a.c0, s);
    r.c1 = fp_mul(a.c1, s);
    return r;
}
// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2)
ZKLC_HD fp2 fp2_inv(const fp2 &a) {
    fp n = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
    fp2 r;
    r.c0 = fp_mul(a.c0, n);
    r.c1 = fp_mul(fp_neg(a.c1), n);
    return r;
}
// 16 words (A0 then A1, gnark Montgomery) -> internal, LAZY (operands of a multiplication only; see fp_from_gnark)
ZKLC_HD fp2 fp2_from_gnark(const u32 *w) {
    fp2 r;
    r.c0 = fp_from_gnark(w);
    r.c1 = fp_from_gnark(w + 8);
    return r;
}
ZKLC_HD void fp2_to_gnark(u32 *out16, const fp2 &a) {
    fp_to_gnark(out16, a.c0);
    fp_to_gnark(out16 + 8, a.c1);
}
