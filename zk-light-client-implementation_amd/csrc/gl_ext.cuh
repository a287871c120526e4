// Quadratic extension of Goldilocks, F[X]/(X^2 - 7) (gnark-plonky2-verifier/goldilocks/quadratic_extension.go:9-10,59-73),
// and the degree-2 "extension algebra" helpers the plonky2 gates use when they are evaluated over the base field
// (goldilocks/quadratic_extension_algebra.go:47-131).
#pragma once
#include "goldilocks.cuh"

struct gl2 {
    u64 a, b;  // a + b X
};

#if defined(__HIPCC__)
__host__
#endif
ZKLC_HD gl2 gl2_make(u64 a, u64 b) {
    gl2 r;
    r.a = a;
    r.b = b;
    return r;
}
ZKLC_HD gl2 gl2_add(gl2 x, gl2 y) { return gl2_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
ZKLC_HD gl2 gl2_sub(gl2 x, gl2 y) { return gl2_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
ZKLC_HD gl2 gl2_neg(gl2 x) { return gl2_make(gl_neg(x.a), gl_neg(x.b)); }
ZKLC_HD u64 gl_mul7(u64 x) {
    // 7x = 8x - x
    u64 x2 = gl_double(x), x4 = gl_double(x2), x8 = gl_double(x4);
    return gl_sub(x8, x);
}
ZKLC_HD gl2 gl2_mul(gl2 x, gl2 y) {
    u64 aa = gl_mul(x.a, y.a), bb = gl_mul(x.b, y.b);
    // (a0 + a1)(b0 + b1) - aa - bb   (Karatsuba; the sums are reduced first)
    u64 cross = gl_sub(gl_sub(gl_mul(gl_add(x.a, x.b), gl_add(y.a, y.b)), aa), bb);
    return gl2_make(gl_add(aa, gl_mul7(bb)), cross);
}
ZKLC_HD gl2 gl2_sqr(gl2 x) {
    u64 aa = gl_sqr(x.a), bb = gl_sqr(x.b), ab = gl_mul(x.a, x.b);
    return gl2_make(gl_add(aa, gl_mul7(bb)), gl_double(ab));
}
ZKLC_HD gl2 gl2_scale(gl2 x, u64 s) { return gl2_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
ZKLC_HD gl2 gl2_add_base(gl2 x, u64 s) { return gl2_make(gl_add(x.a, s), x.b); }
ZKLC_HD gl2 gl2_inv(gl2 x) {
    // 1 / (a + bX) = (a - bX) / (a^2 - 7 b^2)
    u64 d = gl_inv(gl_sub(gl_sqr(x.a), gl_mul7(gl_sqr(x.b))));
    return gl2_make(gl_mul(x.a, d), gl_mul(gl_neg(x.b), d));
}
ZKLC_HD gl2 gl2_pow(gl2 x, u64 e) {
    gl2 r = gl2_make(1, 0);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    while (e) {
        if (e & 1) r = gl2_mul(r, x);
        x = gl2_sqr(x);
        e >>= 1;
    }
    return r;
}
ZKLC_HD bool gl2_eq(gl2 x, gl2 y) { return x.a == y.a && x.b == y.b; }
