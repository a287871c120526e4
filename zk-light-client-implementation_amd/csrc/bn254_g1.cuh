// BN254 G1 (y^2 = x^3 + 3) in extended Jacobian coordinates (X, Y, ZZ, ZZZ):
// x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.  Same coordinate system gnark-crypto uses for its
// MSM buckets (g1JacExtended; un-vendored, gnark-plonky2-verifier/go.mod:9); call site
// `groth16.Prove` -> `MultiExp`, gnark-plonky2-verifier/cmd/web-api.go:77.
// Infinity is ZZ = 0.  Formulas: EFD madd-2008-s / add-2008-s / dbl-2008-s-1, with the
// exceptional cases (P = Q, P = -Q, infinity) handled explicitly -- the adversarial MSM
// inputs of SURVEY 8(d) ("all points equal") reach them.
#pragma once
#include "bn254_fp.cuh"

struct g1_aff {  // internal Montgomery limbs (possibly lazy, see fp_from_gnark); inf = 1: point at infinity
    fp x, y;
    u32 inf;
};
struct g1_xyzz {
    fp X, Y, ZZ, ZZZ;
};

ZKLC_HD g1_xyzz g1_infinity() {
    g1_xyzz r;
    r.X = fp_zero();
    r.Y = fp_zero();
    r.ZZ = fp_zero();
    r.ZZZ = fp_zero();
    return r;
}
ZKLC_HD u32 g1_is_inf(const g1_xyzz &p) { return fp_is_zero(p.ZZ); }

ZKLC_HD g1_xyzz g1_from_affine(const g1_aff &a) {
    const fp one = FP_ONE;
    g1_xyzz r;
    r.X = fp_reduce(a.x);
    r.Y = fp_reduce(a.y);
    r.ZZ = a.inf ? fp_zero() : one;
    r.ZZZ = r.ZZ;
    return r;
}

// dbl-2008-s-1
ZKLC_HD g1_xyzz g1_double(const g1_xyzz &p) {
    fp U = fp_dbl(p.Y);
    fp V = fp_sqr(U);
    fp W = fp_mul(U, V);
    fp S = fp_mul(p.X, V);
    fp XX = fp_sqr(p.X);
    fp M = fp_add(fp_dbl(XX), XX);  // 3 X^2 (a = 0)
    g1_xyzz r;
    r.X = fp_sub(fp_sqr(M), fp_dbl(S));
    r.Y = fp_sub(fp_mul(M, fp_sub(S, r.X)), fp_mul(W, p.Y));
    r.ZZ = fp_mul(V, p.ZZ);
    r.ZZZ = fp_mul(W, p.ZZZ);
    return r;  // doubling infinity (ZZ = 0) gives ZZ = 0 again; a 2-torsion point does not exist on this curve
}

// double of an affine point (mdbl-2008-s-1)
ZKLC_HD g1_xyzz g1_double_affine(const fp &x, const fp &y) {
    fp U = fp_dbl(y);
    g1_xyzz r;
    r.ZZ = fp_sqr(U);
    r.ZZZ = fp_mul(U, r.ZZ);
    fp S = fp_mul(x, r.ZZ);
    fp XX = fp_sqr(x);
    fp M = fp_add(fp_dbl(XX), XX);
    r.X = fp_sub(fp_sqr(M), fp_dbl(S));
    r.Y = fp_sub(fp_mul(M, fp_sub(S, r.X)), fp_mul(r.ZZZ, y));
    return r;
}

// p + (x2, y2) with (x2, y2) affine and finite (lazy fp_from_gnark values allowed);
// neg = 1 adds (x2, -y2)
ZKLC_HD g1_xyzz g1_add_affine(const g1_xyzz &p, const fp &x2, const fp &y2in, u32 neg) {
    fp y2 = fp_select(y2in, fp_neg(y2in), neg);
    if (g1_is_inf(p)) {
        const fp one = FP_ONE;
        g1_xyzz r;
        r.X = fp_reduce(x2);
        r.Y = fp_reduce(y2);
        r.ZZ = one;
        r.ZZZ = one;
        return r;
    }
    fp U2 = fp_mul(x2, p.ZZ);
    fp S2 = fp_mul(y2, p.ZZZ);
    fp Pp = fp_sub(U2, p.X);
    fp R = fp_sub(S2, p.Y);
    if (fp_is_zero(Pp)) {                      // same x: P = +-Q
        if (fp_is_zero(R)) return g1_double_affine(fp_reduce(x2), fp_reduce(y2));
        return g1_infinity();
    }
    fp PP = fp_sqr(Pp);
    fp PPP = fp_mul(Pp, PP);
    fp Q = fp_mul(p.X, PP);
    g1_xyzz r;
    r.X = fp_sub(fp_sub(fp_sqr(R), PPP), fp_dbl(Q));
    r.Y = fp_sub(fp_mul(R, fp_sub(Q, r.X)), fp_mul(p.Y, PPP));
    r.ZZ = fp_mul(p.ZZ, PP);
    r.ZZZ = fp_mul(p.ZZZ, PPP);
    return r;
}

// add-2008-s, general
ZKLC_HD g1_xyzz g1_add(const g1_xyzz &p, const g1_xyzz &q) {
    if (g1_is_inf(p)) return q;
    if (g1_is_inf(q)) return p;
    fp U1 = fp_mul(p.X, q.ZZ);
    fp U2 = fp_mul(q.X, p.ZZ);
    fp S1 = fp_mul(p.Y, q.ZZZ);
    fp S2 = fp_mul(q.Y, p.ZZZ);
    fp Pp = fp_sub(U2, U1);
    fp R = fp_sub(S2, S1);
    if (fp_is_zero(Pp)) {
        if (fp_is_zero(R)) return g1_double(p);
        return g1_infinity();
    }
    fp PP = fp_sqr(Pp);
    fp PPP = fp_mul(Pp, PP);
    fp Q = fp_mul(U1, PP);
    g1_xyzz r;
    r.X = fp_sub(fp_sub(fp_sqr(R), PPP), fp_dbl(Q));
    r.Y = fp_sub(fp_mul(R, fp_sub(Q, r.X)), fp_mul(S1, PPP));
    r.ZZ = fp_mul(fp_mul(p.ZZ, q.ZZ), PP);
    r.ZZZ = fp_mul(fp_mul(p.ZZZ, q.ZZZ), PPP);
    return r;
}

// affine (x, y) = (X / ZZ, Y / ZZZ) in gnark Montgomery words (16 words); returns 1 for infinity (then zeros)
ZKLC_HD u32 g1_to_affine_gnark(u32 *out16, const g1_xyzz &p) {
    if (g1_is_inf(p)) {
#pragma unroll
        for (int i = 0; i < 16; i++) out16[i] = 0;
        return 1;
    }
    // one inversion: (ZZ * ZZZ)^-1 -> 1/ZZ = inv * ZZZ, 1/ZZZ = inv * ZZ
    fp inv = fp_inv(fp_mul(p.ZZ, p.ZZZ));
    fp x = fp_mul(p.X, fp_mul(inv, p.ZZZ));
    fp y = fp_mul(p.Y, fp_mul(inv, p.ZZ));
    fp_to_gnark(out16, x);
    fp_to_gnark(out16 + 8, y);
    return 0;
}
