// BN254 optimal-ate pairing: Fp6 / Fp12 tower, Miller loop in homogeneous projective coordinates, final exponentiation.
//
// Replaces gnark-crypto's `bn254.PairingCheck` (un-vendored) behind `groth16.Verify`
// (gnark-plonky2-verifier/cmd/web-api.go:84) and the EVM pairing precompile the reference's Solidity verifier calls
// (contracts/hardhat/contracts/Verifier.sol:503-548).  Tower as in gnark-crypto: Fp2 = Fp[u]/(u^2+1),
// Fp6 = Fp2[v]/(v^3 - (9+u)), Fp12 = Fp6[w]/(w^2 - v); D-type twist; loop 6x+2, x = 4965661367192848881 (the `t` of
// Verifier.sol:29-33).  Oracle: oracle/bn254_pairing.py (pinned by the reference's Groth16 known-answer proof).
//
// Field elements are the lazy ten-limb Fp of bn254_fp.cuh.  Invariant of every f6_/f12_ function here: coordinates of
// inputs and outputs are "normal" (|value| <= 2p, limbs in [0, 2^26) except the signed top limb); sums and multiples
// of xi = 9 + u are brought back with fp_wred, a 60-instruction weak reduction (subtract round(top/p_top) * p), so
// no operand of a multiplication ever exceeds 8p (bn254_fp.cuh: legal up to 16p, limbs < 2^30).
// The Miller lines are scaled by Fp2 factors (projective coordinates); the final exponentiation removes them, so the
// GT element equals the oracle's affine-coordinate result.  Hard part: plain exponentiation by (p^4 - p^2 + 1)/r.
#pragma once
#include "bn254_fp2.cuh"

#if defined(__HIPCC__)
#define ZKLC_CONST_ARRAY_PAIRING __device__ __constant__ const
// the tower multiplications are real device functions: inlining every call site of the Miller loop and the final
// exponentiation would be millions of instructions
#define ZKLC_TOWER __device__ __noinline__
#else
#define ZKLC_CONST_ARRAY_PAIRING static const
#define ZKLC_TOWER static
#endif

// |a| <= 64p, limbs < 2^30  ->  same element with |value| < 2p, limbs 0..8 in [0, 2^26)
ZKLC_HD fp fp_wred(const fp &a) {
    const i32 Pm[10] = FP_P26;
    i64 t[10];
    i64 c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        i64 x = (i64)a.v[i] + c;
        t[i] = x & 0x3ffffff;
        c = x >> 26;
    }
    t[9] = (i64)a.v[9] + c;
    // quotient estimate from the top limb (p / 2^234 = 792851.6...)
    i32 q = (i32)(((double)t[9]) * (1.0 / 792851.6) + (t[9] >= 0 ? 0.5 : -0.5));
    fp r;
    c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        i64 x = t[i] - (i64)q * Pm[i] + c;
        r.v[i] = (i32)(x & 0x3ffffff);
        c = x >> 26;
    }
    r.v[9] = (i32)(t[9] - (i64)q * Pm[9] + c);
    return r;
}
ZKLC_HD fp2 fp2_wred(const fp2 &a) {
    fp2 r;
    r.c0 = fp_wred(a.c0);
    r.c1 = fp_wred(a.c1);
    return r;
}
// (9 + u) * a, reduced.  a normal.
ZKLC_HD fp2 fp2_mul_xi(const fp2 &a) {
    fp2 r;
    fp a0_8 = fp_dbl(fp_dbl(fp_dbl(a.c0))), a1_8 = fp_dbl(fp_dbl(fp_dbl(a.c1)));
    r.c0 = fp_wred(fp_sub(fp_add(a0_8, a.c0), a.c1));   // 9 a0 - a1
    r.c1 = fp_wred(fp_add(fp_add(a1_8, a.c1), a.c0));   // 9 a1 + a0
    return r;
}
ZKLC_HD u32 fp2_eq(const fp2 &a, const fp2 &b) { return fp2_is_zero(fp2_sub(a, b)); }

// ---------------------------------------------------------------- Fp6
struct fp6 {
    fp2 b0, b1, b2;
};
ZKLC_HD fp6 f6_zero() {
    fp6 r;
    r.b0 = r.b1 = r.b2 = fp2_zero();
    return r;
}
ZKLC_HD fp6 f6_one() {
    fp6 r = f6_zero();
    r.b0 = fp2_one();
    return r;
}
ZKLC_HD fp6 f6_add(const fp6 &a, const fp6 &b) {
    fp6 r;
    r.b0 = fp2_wred(fp2_add(a.b0, b.b0));
    r.b1 = fp2_wred(fp2_add(a.b1, b.b1));
    r.b2 = fp2_wred(fp2_add(a.b2, b.b2));
    return r;
}
ZKLC_HD fp6 f6_sub(const fp6 &a, const fp6 &b) {
    fp6 r;
    r.b0 = fp2_wred(fp2_sub(a.b0, b.b0));
    r.b1 = fp2_wred(fp2_sub(a.b1, b.b1));
    r.b2 = fp2_wred(fp2_sub(a.b2, b.b2));
    return r;
}
ZKLC_HD fp6 f6_neg(const fp6 &a) {
    fp6 r;
    r.b0 = fp2_neg(a.b0);
    r.b1 = fp2_neg(a.b1);
    r.b2 = fp2_neg(a.b2);
    return r;
}
// Karatsuba: 6 Fp2 multiplications
ZKLC_TOWER fp6 f6_mul(const fp6 &a, const fp6 &b) {
    fp2 t0 = fp2_mul(a.b0, b.b0), t1 = fp2_mul(a.b1, b.b1), t2 = fp2_mul(a.b2, b.b2);
    fp2 s12 = fp2_mul(fp2_add(a.b1, a.b2), fp2_add(b.b1, b.b2));
    fp2 s01 = fp2_mul(fp2_add(a.b0, a.b1), fp2_add(b.b0, b.b1));
    fp2 s02 = fp2_mul(fp2_add(a.b0, a.b2), fp2_add(b.b0, b.b2));
    fp6 r;
    r.b0 = fp2_wred(fp2_add(t0, fp2_mul_xi(fp2_wred(fp2_sub(fp2_sub(s12, t1), t2)))));
    r.b1 = fp2_wred(fp2_add(fp2_sub(fp2_sub(s01, t0), t1), fp2_mul_xi(t2)));
    r.b2 = fp2_wred(fp2_add(fp2_sub(fp2_sub(s02, t0), t2), t1));
    return r;
}
ZKLC_HD fp6 f6_mul_by_v(const fp6 &a) {
    fp6 r;
    r.b0 = fp2_mul_xi(a.b2);
    r.b1 = a.b0;
    r.b2 = a.b1;
    return r;
}
ZKLC_HD fp6 f6_scale(const fp6 &a, const fp2 &s) {
    fp6 r;
    r.b0 = fp2_mul(a.b0, s);
    r.b1 = fp2_mul(a.b1, s);
    r.b2 = fp2_mul(a.b2, s);
    return r;
}
// a * (c0 + c1 v)
ZKLC_TOWER fp6 f6_mul_01(const fp6 &a, const fp2 &c0, const fp2 &c1) {
    fp2 t0 = fp2_mul(a.b0, c0), t1 = fp2_mul(a.b1, c1);
    fp6 r;
    r.b0 = fp2_wred(fp2_add(t0, fp2_mul_xi(fp2_mul(a.b2, c1))));
    r.b1 = fp2_wred(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.b0, a.b1), fp2_add(c0, c1)), t0), t1));
    r.b2 = fp2_wred(fp2_add(fp2_mul(a.b2, c0), t1));
    return r;
}
ZKLC_TOWER fp6 f6_inv(const fp6 &a) {
    fp2 t0 = fp2_wred(fp2_sub(fp2_sqr(a.b0), fp2_mul_xi(fp2_mul(a.b1, a.b2))));
    fp2 t1 = fp2_wred(fp2_sub(fp2_mul_xi(fp2_sqr(a.b2)), fp2_mul(a.b0, a.b1)));
    fp2 t2 = fp2_wred(fp2_sub(fp2_sqr(a.b1), fp2_mul(a.b0, a.b2)));
    fp2 d = fp2_wred(fp2_add(fp2_mul(a.b0, t0), fp2_mul_xi(fp2_wred(fp2_add(fp2_mul(a.b2, t1), fp2_mul(a.b1, t2))))));
    fp2 di = fp2_inv(d);
    fp6 r;
    r.b0 = fp2_mul(t0, di);
    r.b1 = fp2_mul(t1, di);
    r.b2 = fp2_mul(t2, di);
    return r;
}

// ---------------------------------------------------------------- Fp12
struct fp12 {
    fp6 c0, c1;
};
ZKLC_HD fp12 f12_one() {
    fp12 r;
    r.c0 = f6_one();
    r.c1 = f6_zero();
    return r;
}
ZKLC_TOWER fp12 f12_mul(const fp12 &a, const fp12 &b) {
    fp6 t0 = f6_mul(a.c0, b.c0), t1 = f6_mul(a.c1, b.c1);
    fp12 r;
    r.c1 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1)), t0), t1);
    r.c0 = f6_add(t0, f6_mul_by_v(t1));
    return r;
}
// complex squaring: c0 = (a0 + a1)(a0 + v a1) - (1 + v) a0 a1, c1 = 2 a0 a1
ZKLC_TOWER fp12 f12_sqr(const fp12 &a) {
    fp6 m = f6_mul(a.c0, a.c1);
    fp6 s = f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_by_v(a.c1)));
    fp12 r;
    r.c0 = f6_sub(f6_sub(s, m), f6_mul_by_v(m));
    r.c1 = f6_add(m, m);
    return r;
}
ZKLC_HD fp12 f12_conj(const fp12 &a) {
    fp12 r;
    r.c0 = a.c0;
    r.c1 = f6_neg(a.c1);
    return r;
}
ZKLC_TOWER fp12 f12_inv(const fp12 &a) {
    fp6 d = f6_inv(f6_sub(f6_mul(a.c0, a.c0), f6_mul_by_v(f6_mul(a.c1, a.c1))));
    fp12 r;
    r.c0 = f6_mul(a.c0, d);
    r.c1 = f6_neg(f6_mul(a.c1, d));
    return r;
}
// f * (a + b w + c v w): the sparse line value, a, b, c in Fp2
ZKLC_TOWER fp12 f12_mul_line(const fp12 &f, const fp2 &a, const fp2 &b, const fp2 &c) {
    fp6 t0 = f6_scale(f.c0, a);
    fp6 t1 = f6_mul_01(f.c1, b, c);
    fp12 r;
    r.c1 = f6_sub(f6_sub(f6_mul_01(f6_add(f.c0, f.c1), fp2_add(a, b), c), t0), t1);
    r.c0 = f6_add(t0, f6_mul_by_v(t1));
    return r;
}

// xi^(i (p^k - 1) / 6), i = 1..5, k = 1, 2 (internal Montgomery limbs; generated by the script quoted in
// tools/gen_constants.py --pairing; checked against oracle/bn254_pairing.py in tests/test_hostsim_pairing.py)
ZKLC_CONST_ARRAY_PAIRING fp2 BN_GAMMA[10] = {
    /* gamma_1,1 */ {{{21270640, 43669708, 59420571, 14805997, 10924503, 36209183, 32333552, 43975791, 55681523, 773429}}, {{35775677, 34173090, 34570072, 28929684, 17525013, 27096614, 48093515, 13710501, 19714721, 401402}}},
    /* gamma_1,2 */ {{{38979784, 64230244, 51255142, 66935903, 42724807, 28669321, 5131235, 20309878, 32116372, 299606}}, {{13145628, 14169593, 22511980, 2025678, 59225784, 50274257, 48203033, 15276842, 38951387, 555091}}},
    /* gamma_1,3 */ {{{26182460, 30403611, 48363427, 24426392, 10026519, 17646722, 27094552, 3111387, 42451388, 239831}}, {{17932942, 606602, 10489147, 9384144, 36826109, 11975780, 8398858, 11139828, 31460665, 572781}}},
    /* gamma_1,4 */ {{{4258863, 40424982, 22915347, 14835858, 53946579, 39203778, 16822642, 58103548, 12030369, 184042}}, {{61912380, 38801573, 58093765, 36010944, 51552667, 24330700, 5423403, 37190496, 7375051, 57151}}},
    /* gamma_1,5 */ {{{61017798, 13802176, 51827428, 44568487, 9894594, 13591319, 13387539, 51143528, 2057182, 136318}}, {{60367719, 57096022, 41067753, 1462599, 7405059, 58669173, 23244912, 51178937, 43526976, 462728}}},
    /* gamma_2,1 */ {{{52740569, 31264333, 46200523, 47483678, 24435702, 64272972, 51254852, 54058150, 65588307, 297769}}, {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}},
    /* gamma_2,2 */ {{{10803820, 25031622, 47138360, 66273855, 12768344, 64049169, 25194452, 66078065, 889293, 581293}}, {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}},
    /* gamma_2,3 */ {{{25172115, 60876152, 937836, 18790177, 55441506, 66885060, 41048463, 12019914, 2409850, 283523}}, {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}},
    /* gamma_2,4 */ {{{22559598, 38139752, 31972598, 57743016, 2270579, 9149387, 32916771, 55036474, 42603741, 495081}}, {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}},
    /* gamma_2,5 */ {{{64496347, 44372463, 31034761, 38952839, 13937937, 9373190, 58977171, 43016559, 40193891, 211558}}, {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}},
};
// k-th power Frobenius (k = 1 or 2): conjugate (k odd) and scale the coefficient of v^i w^j by gamma_{k, 2i + j}
ZKLC_TOWER fp12 f12_frobenius(const fp12 &a, int k) {
    const fp2 *g = BN_GAMMA + (k == 1 ? 0 : 5);
    fp2 x[6] = {a.c0.b0, a.c1.b0, a.c0.b1, a.c1.b1, a.c0.b2, a.c1.b2};  // coefficients of w^0 .. w^5
    if (k == 1)
        for (int i = 0; i < 6; i++) x[i] = fp2_conj(x[i]);
    for (int i = 1; i < 6; i++) x[i] = fp2_mul(x[i], g[i - 1]);
    fp12 r;
    r.c0.b0 = x[0];
    r.c1.b0 = x[1];
    r.c0.b1 = x[2];
    r.c1.b1 = x[3];
    r.c0.b2 = x[4];
    r.c1.b2 = x[5];
    return r;
}

// ---------------------------------------------------------------- Miller loop
struct g2_proj {
    fp2 X, Y, Z;
};
// T <- 2T; line through T tangent, evaluated at P = (xp, yp), scaled by 2 Y Z^2:
//   a = 2 Y Z^2 yp, b = -3 X^2 Z xp, c = 3 X^3 - 2 Y^2 Z
ZKLC_TOWER void bn_double_step(g2_proj &T, const fp &xp, const fp &yp, fp2 &a, fp2 &b, fp2 &c) {
    fp2 XX = fp2_sqr(T.X), YY = fp2_sqr(T.Y), ZZ = fp2_sqr(T.Z);
    fp2 W = fp2_wred(fp2_add(fp2_dbl(XX), XX));            // 3 X^2
    fp2 S = fp2_mul(T.Y, T.Z);                              // Y Z
    a = fp2_mul_fp(fp2_wred(fp2_dbl(fp2_mul(T.Y, ZZ))), yp);
    b = fp2_neg(fp2_mul_fp(fp2_mul(W, T.Z), xp));
    c = fp2_wred(fp2_sub(fp2_mul(W, T.X), fp2_dbl(fp2_mul(YY, T.Z))));
    // homogeneous doubling (a = 0): Bq = X Y S, H = W^2 - 8 Bq, X3 = 2 H S, Y3 = W (4 Bq - H) - 8 Y^2 S^2, Z3 = 8 S^3
    fp2 Bq = fp2_mul(fp2_mul(T.X, T.Y), S);
    fp2 B4 = fp2_wred(fp2_dbl(fp2_dbl(Bq)));
    fp2 H = fp2_wred(fp2_sub(fp2_sqr(W), fp2_dbl(B4)));
    fp2 SS = fp2_sqr(S);
    fp2 X3 = fp2_wred(fp2_dbl(fp2_mul(H, S)));
    fp2 Y3 = fp2_wred(fp2_sub(fp2_mul(W, fp2_wred(fp2_sub(B4, H))), fp2_wred(fp2_dbl(fp2_dbl(fp2_dbl(fp2_mul(YY, SS)))))));
    fp2 Z3 = fp2_wred(fp2_dbl(fp2_dbl(fp2_dbl(fp2_mul(S, SS)))));
    T.X = X3;
    T.Y = Y3;
    T.Z = Z3;
}
// T <- T + Q (Q affine); line through T and Q at P scaled by mu = x2 Z - X:
//   a = mu yp, b = -theta xp, c = theta x2 - mu y2,  theta = y2 Z - Y
ZKLC_TOWER void bn_add_step(g2_proj &T, const fp2 &x2, const fp2 &y2, const fp &xp, const fp &yp, fp2 &a, fp2 &b, fp2 &c) {
    fp2 theta = fp2_wred(fp2_sub(fp2_mul(y2, T.Z), T.Y));
    fp2 mu = fp2_wred(fp2_sub(fp2_mul(x2, T.Z), T.X));
    a = fp2_mul_fp(mu, yp);
    b = fp2_neg(fp2_mul_fp(theta, xp));
    c = fp2_wred(fp2_sub(fp2_mul(theta, x2), fp2_mul(mu, y2)));
    fp2 vv = fp2_sqr(mu), vvv = fp2_mul(mu, vv);
    fp2 Rr = fp2_mul(vv, T.X);
    fp2 A = fp2_wred(fp2_sub(fp2_sub(fp2_mul(fp2_sqr(theta), T.Z), vvv), fp2_dbl(Rr)));
    fp2 X3 = fp2_mul(mu, A);
    fp2 Y3 = fp2_wred(fp2_sub(fp2_mul(theta, fp2_wred(fp2_sub(Rr, A))), fp2_mul(vvv, T.Y)));
    fp2 Z3 = fp2_mul(vvv, T.Z);
    T.X = X3;
    T.Y = Y3;
    T.Z = Z3;
}

// f *= f_{6x+2,Q}(P) l(pi Q) l(-pi^2 Q); P = (xp, yp) and Q = (xq, yq) affine, normal; skipped when either is infinity
ZKLC_HD void bn_miller_loop(fp12 &f, const fp &xp, const fp &yp, const fp2 &xq, const fp2 &yq) {
    // 6x + 2 = 0x19d797039be763ba8 (65 bits): bits below the leading one, most significant first
    const u64 LOOP_LO = 0x9d797039be763ba8ULL;  // low 64 bits; the 65th bit is the leading one
    g2_proj T;
    T.X = xq;
    T.Y = yq;
    T.Z = fp2_one();
    fp12 acc = f12_one();
    fp2 a, b, c;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int i = 63; i >= 0; i--) {
        bn_double_step(T, xp, yp, a, b, c);
        acc = f12_mul_line(f12_sqr(acc), a, b, c);
        if ((LOOP_LO >> i) & 1) {
            bn_add_step(T, xq, yq, xp, yp, a, b, c);
            acc = f12_mul_line(acc, a, b, c);
        }
    }
    // Q1 = pi(Q), Q2 = -pi^2(Q)
    fp2 x1 = fp2_mul(fp2_conj(xq), BN_GAMMA[1]), y1 = fp2_mul(fp2_conj(yq), BN_GAMMA[2]);
    fp2 x2 = fp2_mul(xq, BN_GAMMA[6]), y2 = fp2_neg(fp2_mul(yq, BN_GAMMA[7]));
    bn_add_step(T, x1, y1, xp, yp, a, b, c);
    acc = f12_mul_line(acc, a, b, c);
    bn_add_step(T, x2, y2, xp, yp, a, b, c);
    acc = f12_mul_line(acc, a, b, c);
    f = f12_mul(f, acc);
}

// squaring in the cyclotomic subgroup (where f lives after the easy part of the final exponentiation): Granger-Scott,
// https://eprint.iacr.org/2009/565 section 3.2, on the tower Fp12 = Fp6[w] / (w^2 - v), Fp6 = Fp2[v] / (v^3 - xi) -- nine Fp2
// squarings instead of the 18 Fp2 products of the complex squaring (the form gnark-crypto's E12.CyclotomicSquare takes, un-vendored:
// gnark-plonky2-verifier/go.mod:9).  x = (x0 .. x5) = (c0.b0, c0.b1, c0.b2, c1.b0, c1.b1, c1.b2):
//   (3 x4^2 xi + 3 x0^2 - 2 x0, 3 x2^2 xi + 3 x3^2 - 2 x1, 3 x5^2 xi + 3 x1^2 - 2 x2, 6 x1 x5 xi + 2 x3, 6 x0 x4 + 2 x4, 6 x2 x3 + 2 x5)
ZKLC_TOWER fp12 f12_cyclo_sqr(const fp12 &a) {
    fp2 x0 = fp2_wred(a.c0.b0), x1 = fp2_wred(a.c0.b1), x2 = fp2_wred(a.c0.b2);
    fp2 x3 = fp2_wred(a.c1.b0), x4 = fp2_wred(a.c1.b1), x5 = fp2_wred(a.c1.b2);
    fp2 t0 = fp2_sqr(x4), t1 = fp2_sqr(x0);
    fp2 t6 = fp2_wred(fp2_sub(fp2_sub(fp2_sqr(fp2_add(x4, x0)), t0), t1));            // 2 x4 x0
    fp2 t2 = fp2_sqr(x2), t3 = fp2_sqr(x3);
    fp2 t7 = fp2_wred(fp2_sub(fp2_sub(fp2_sqr(fp2_add(x2, x3)), t2), t3));            // 2 x2 x3
    fp2 t4 = fp2_sqr(x5), t5 = fp2_sqr(x1);
    fp2 t8 = fp2_mul_xi(fp2_wred(fp2_sub(fp2_sub(fp2_sqr(fp2_add(x5, x1)), t4), t5)));  // 2 x5 x1 xi
    t0 = fp2_wred(fp2_add(fp2_mul_xi(fp2_wred(t0)), t1));                              // x4^2 xi + x0^2
    t2 = fp2_wred(fp2_add(fp2_mul_xi(fp2_wred(t2)), t3));                              // x2^2 xi + x3^2
    t4 = fp2_wred(fp2_add(fp2_mul_xi(fp2_wred(t4)), t5));                              // x5^2 xi + x1^2
    fp12 r;
    r.c0.b0 = fp2_wred(fp2_add(fp2_dbl(fp2_sub(t0, x0)), t0));
    r.c0.b1 = fp2_wred(fp2_add(fp2_dbl(fp2_sub(t2, x1)), t2));
    r.c0.b2 = fp2_wred(fp2_add(fp2_dbl(fp2_sub(t4, x2)), t4));
    r.c1.b0 = fp2_wred(fp2_add(fp2_dbl(fp2_add(t8, x3)), t8));
    r.c1.b1 = fp2_wred(fp2_add(fp2_dbl(fp2_add(t6, x4)), t6));
    r.c1.b2 = fp2_wred(fp2_add(fp2_dbl(fp2_add(t7, x5)), t7));
    return r;
}
// f^x for the BN parameter x = 4965661367192848881 (63 bits, Hamming weight 28; the `t` of Verifier.sol:29-33), f cyclotomic
ZKLC_TOWER fp12 f12_pow_x(const fp12 &f) {
    const u64 X = 0x44E992B44A6909F1ULL;
    fp12 r = f;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int i = 61; i >= 0; i--) {
        r = f12_cyclo_sqr(r);
        if ((X >> i) & 1) r = f12_mul(r, f);
    }
    return r;
}

// f^((p^12 - 1) / r), the EXACT exponent (the value gnark-crypto's `FinalExponentiation` returns up to its fixed cofactor power is
// not needed here: the oracle and the contract only ask whether the product is one, and the tests compare GT values with the
// oracle's plain exponentiation).  Easy part (p^6 - 1)(p^2 + 1); hard part (p^4 - p^2 + 1) / r = p^3 + (6x^2 + 1) p^2 +
// (-36x^3 - 18x^2 - 12x + 1) p + (-36x^3 - 30x^2 - 18x - 2) (Scott, Benger, Charlemagne, Dominguez Perez, Kachisa: "On the final
// exponentiation for calculating pairings on ordinary elliptic curves"): three exponentiations by x with cyclotomic squarings,
// Frobenius maps, and the vectorial addition chain y0 y1^2 y2^6 y3^12 y4^18 y5^30 y6^36 -- ~190 cyclotomic squarings and ~95
// products instead of the 760 squarings and ~380 products of square-and-multiply over the 761-bit exponent (round 1).
ZKLC_HD fp12 bn_final_exponentiation(const fp12 &f_in) {
    fp12 f = f12_mul(f12_conj(f_in), f12_inv(f_in));
    f = f12_mul(f12_frobenius(f, 2), f);
    fp12 fx = f12_pow_x(f), fx2 = f12_pow_x(fx), fx3 = f12_pow_x(fx2);
    fp12 fp1 = f12_frobenius(f, 1), fp2_ = f12_frobenius(f, 2);
    fp12 y0 = f12_mul(f12_mul(fp1, fp2_), f12_frobenius(fp2_, 1));                      // f^p f^(p^2) f^(p^3)
    fp12 y1 = f12_conj(f);
    fp12 y2 = f12_frobenius(fx2, 2);
    fp12 y3 = f12_conj(f12_frobenius(fx, 1));
    fp12 y4 = f12_conj(f12_mul(fx, f12_frobenius(fx2, 1)));
    fp12 y5 = f12_conj(fx2);
    fp12 y6 = f12_conj(f12_mul(fx3, f12_frobenius(fx3, 1)));
    fp12 t0 = f12_cyclo_sqr(y6);
    t0 = f12_mul(t0, y4);
    t0 = f12_mul(t0, y5);
    fp12 t1 = f12_mul(y3, y5);
    t1 = f12_mul(t1, t0);
    t0 = f12_mul(t0, y2);
    t1 = f12_cyclo_sqr(t1);
    t1 = f12_mul(t1, t0);
    t1 = f12_cyclo_sqr(t1);
    t0 = f12_mul(t1, y1);
    t1 = f12_mul(t1, y0);
    t0 = f12_cyclo_sqr(t0);
    return f12_mul(t0, t1);
}

ZKLC_HD u32 f12_is_one(const fp12 &a) {
    u32 ok = fp2_eq(a.c0.b0, fp2_one());
    ok &= fp2_is_zero(a.c0.b1) & fp2_is_zero(a.c0.b2) & fp2_is_zero(a.c1.b0) & fp2_is_zero(a.c1.b1) & fp2_is_zero(a.c1.b2);
    return ok;
}
// 12 Fp coefficients (tower order c0.b0.a0, c0.b0.a1, c0.b1.a0, ...) in gnark Montgomery words: 96 u32
ZKLC_HD void f12_to_gnark(u32 *out, const fp12 &a) {
    const fp2 *x[6] = {&a.c0.b0, &a.c0.b1, &a.c0.b2, &a.c1.b0, &a.c1.b1, &a.c1.b2};
    for (int i = 0; i < 6; i++) fp2_to_gnark(out + 16 * i, *x[i]);
}
