// edwards25519 group operations in extended coordinates (X:Y:Z:T), a = -1.
//
// Replaces the reference's native curve code for the signature pre-check:
// affine/projective add + double of crypto/plonky2_ed25519/src/curve/curve_adds.rs:8-60
// and curve/curve_types.rs:171-223, the bit-serial `mul_naive`
// (curve/ed25519.rs:55-72), and the dalek `decompress` the reference delegates
// to (curve/eddsa.rs:19-31).  The formulas are the complete unified
// Hisil-Wong-Carter-Dawson ones, so every input on the curve (including
// small-order points) follows the group law exactly.
#pragma once
#include "fe25519.cuh"

struct ge_p3 {  // extended: x = X/Z, y = Y/Z, T = XY/Z
    fe X, Y, Z, T;
};
struct ge_cached {  // (Y+X, Y-X, Z, 2dT)
    fe YpX, YmX, Z, T2d;
};
struct ge_niels {  // affine cached: (y+x, y-x, 2dxy), Z = 1
    fe ypx, ymx, xy2d;
};

ZKLC_HD ge_p3 ge_identity() {
    ge_p3 r;
    r.X = fe_zero();
    r.Y = fe_one();
    r.Z = fe_one();
    r.T = fe_zero();
    return r;
}

ZKLC_HD ge_cached ge_to_cached(const ge_p3 &p) {
    const fe d2 = FE_2D;
    ge_cached c;
    c.YpX = fe_add(p.Y, p.X);
    c.YmX = fe_sub(p.Y, p.X);
    c.Z = p.Z;
    c.T2d = fe_mul(p.T, d2);
    return c;
}

// dbl-2008-hwcd.  want_t (wave-uniform) = the next operation is an addition and
// needs T; a following doubling does not.
ZKLC_HD ge_p3 ge_double(const ge_p3 &p, bool want_t) {
    fe A = fe_sqr(p.X);
    fe B = fe_sqr(p.Y);
    fe C = fe_sqr2(p.Z);                                   // 2 Z^2, reduced
    fe E = fe_sub(fe_sub(fe_sqr(fe_add(p.X, p.Y)), A), B);  // 2XY
    fe G = fe_sub(B, A);                                   // D + B with D = -A
    fe F = fe_sub(G, C);
    fe H = fe_sub(fe_neg(A), B);  // D - B
    ge_p3 r;
    r.X = fe_mul(E, F);
    r.Y = fe_mul(G, H);
    r.Z = fe_mul(F, G);
    if (want_t) r.T = fe_mul(E, H);
    else r.T = fe_zero();
    return r;
}

// add-2008-hwcd-3 with a cached second operand; neg=1 adds -q
ZKLC_HD ge_p3 ge_add_cached(const ge_p3 &p, const ge_cached &q, u32 neg, bool want_t) {
    fe qa = fe_select(q.YmX, q.YpX, neg);  // (Y2-X2) or, for -q, (Y2+X2)
    fe qb = fe_select(q.YpX, q.YmX, neg);
    fe A = fe_mul(fe_sub(p.Y, p.X), qa);
    fe B = fe_mul(fe_add(p.Y, p.X), qb);
    fe C = fe_mul(p.T, q.T2d);
    C = fe_select(C, fe_neg(C), neg);
    fe ZZ = fe_mul(p.Z, q.Z);
    fe D = fe_add(ZZ, ZZ);
    fe E = fe_sub(B, A);
    fe F = fe_sub(D, C);
    fe G = fe_add(D, C);
    fe H = fe_add(B, A);
    ge_p3 r;
    r.X = fe_mul(E, F);
    r.Y = fe_mul(G, H);
    r.Z = fe_mul(F, G);
    if (want_t) r.T = fe_mul(E, H);
    else r.T = fe_zero();
    return r;
}

// mixed addition with an affine-niels operand (Z2 = 1); neg=1 adds -q
ZKLC_HD ge_p3 ge_add_niels(const ge_p3 &p, const ge_niels &q, u32 neg, bool want_t) {
    fe qa = fe_select(q.ymx, q.ypx, neg);
    fe qb = fe_select(q.ypx, q.ymx, neg);
    fe A = fe_mul(fe_sub(p.Y, p.X), qa);
    fe B = fe_mul(fe_add(p.Y, p.X), qb);
    fe C = fe_mul(p.T, q.xy2d);
    C = fe_select(C, fe_neg(C), neg);
    fe D = fe_add(p.Z, p.Z);
    fe E = fe_sub(B, A);
    fe F = fe_sub(D, C);
    fe G = fe_add(D, C);
    fe H = fe_add(B, A);
    ge_p3 r;
    r.X = fe_mul(E, F);
    r.Y = fe_mul(G, H);
    r.Z = fe_mul(F, G);
    if (want_t) r.T = fe_mul(E, H);
    else r.T = fe_zero();
    return r;
}

// curve25519-dalek CompressedEdwardsY::decompress on 8 little-endian words.
// Returns 1 on success.  y is NOT checked for canonicity and x = 0 with the
// sign bit set is accepted, exactly as dalek does (SURVEY 9.4 ii).
ZKLC_HD u32 ge_decompress(ge_p3 &r, const u32 *w) {
    const fe d = FE_D;
    const fe sqrtm1 = FE_SQRTM1;
    u32 sign = w[7] >> 31;
    fe y = fe_from_words(w);
    fe yy = fe_sqr(y);
    fe u = fe_sub(yy, fe_one());
    fe v = fe_add(fe_mul(yy, d), fe_one());
    // sqrt_ratio_i(u, v): r = u v^3 (u v^7)^((p-5)/8)
    fe v3 = fe_mul(fe_sqr(v), v);
    fe v7 = fe_mul(fe_sqr(v3), v);
    fe x = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sqr(x));
    u32 correct = fe_eq(check, u);
    u32 flipped = fe_eq(check, fe_neg(u));
    x = fe_select(x, fe_mul(x, sqrtm1), flipped);
    u32 ok = correct | flipped;
    // non-negative root, then apply the sign bit
    u32 isneg = fe_is_negative(x);
    x = fe_select(x, fe_neg(x), isneg ^ sign);
    r.X = x;
    r.Y = y;
    r.Z = fe_one();
    r.T = fe_mul(x, y);
    return ok;
}

// canonical 32-byte encoding as 8 little-endian words
ZKLC_HD void ge_compress(u32 *out, const ge_p3 &p) {
    fe zi = fe_invert(p.Z);
    fe x = fe_mul(p.X, zi);
    u32 s = fe_is_negative(x);
    fe_freeze_words(out, fe_mul(p.Y, zi));
    out[7] |= s << 31;
}

ZKLC_HD ge_niels ge_to_niels(const ge_p3 &p) {  // normalises Z (one inversion)
    const fe d2 = FE_2D;
    fe zi = fe_invert(p.Z);
    fe x = fe_mul(p.X, zi), y = fe_mul(p.Y, zi);
    ge_niels n;
    n.ypx = fe_freeze(fe_add(y, x));
    n.ymx = fe_freeze(fe_sub(y, x));
    n.xy2d = fe_freeze(fe_mul(fe_mul(x, y), d2));
    return n;
}

// base point B (crypto/plonky2_ed25519/src/curve/ed25519.rs:37-51)
#define GE_BASE_X {{52811034, 25909283, 16144682, 17082669, 27570973, 30858332, 40966398, 8378388, 20764389, 8758491}}
#define GE_BASE_Y {{40265304, 26843545, 13421772, 20132659, 26843545, 6710886, 53687091, 13421772, 40265318, 26843545}}

ZKLC_HD ge_p3 ge_base() {
    const fe bx = GE_BASE_X, by = GE_BASE_Y;
    ge_p3 r;
    r.X = bx;
    r.Y = by;
    r.Z = fe_one();
    r.T = fe_mul(bx, by);
    return r;
}
