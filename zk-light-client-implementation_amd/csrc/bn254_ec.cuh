// Short-Weierstrass curves y^2 = x^3 + b (a = 0) in extended Jacobian coordinates (X, Y, ZZ, ZZZ), generic over the
// coordinate field: BN254 G1 over Fp (FpField) and G2 over Fp2 (Fp2Field) share these formulas.
// x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2 -- the bucket coordinates of gnark-crypto's MSM (g1JacExtended / g2JacExtended,
// un-vendored: gnark-plonky2-verifier/go.mod:9; call site `groth16.Prove`, cmd/web-api.go:77).  Infinity is ZZ = 0.
// EFD madd-2008-s / add-2008-s / dbl-2008-s-1; the exceptional cases (P = Q, P = -Q, infinity) are explicit because
// the adversarial MSM inputs of SURVEY 8(d) ("all points equal") reach them.  b never appears (a = 0 formulas).
#pragma once
#include "bn254_fp2.cuh"

struct FpField {
    typedef fp T;
    static constexpr int LIMBS = 10;
    static ZKLC_M T zero() { return fp_zero(); }
    static ZKLC_M T one() {
        const fp o = FP_ONE;
        return o;
    }
    static ZKLC_M T add(const T &a, const T &b) { return fp_add(a, b); }
    static ZKLC_M T sub(const T &a, const T &b) { return fp_sub(a, b); }
    static ZKLC_M T dbl(const T &a) { return fp_dbl(a); }
    static ZKLC_M T neg(const T &a) { return fp_neg(a); }
    static ZKLC_M T mul(const T &a, const T &b) { return fp_mul(a, b); }
    static ZKLC_M T sqr(const T &a) { return fp_sqr(a); }
    static ZKLC_M T inv(const T &a) { return fp_inv(a); }
    static ZKLC_M T reduce(const T &a) { return fp_reduce(a); }
    static ZKLC_M T select(const T &a, const T &b, u32 c) { return fp_select(a, b, c); }
    static ZKLC_M u32 is_zero(const T &a) { return fp_is_zero(a); }
    static ZKLC_M T from_gnark(const u32 *w) { return fp_from_gnark(w); }
    static ZKLC_M void to_gnark(u32 *w, const T &a) { fp_to_gnark(w, a); }
    // packed record form: the canonical value in [0, p) (internal Montgomery domain) as 8 x 32 bits -- 32 bytes instead of 40
    static constexpr int PACKW = 8;
    static ZKLC_M T glv_beta() {           // the cube root of unity of the endomorphism (x, y) -> (beta x, y), internal domain
        const fp b = {{22559598, 38139752, 31972598, 57743016, 2270579, 9149387, 32916771, 55036474, 42603741, 495081}};
        return b;
    }
    static ZKLC_M void pack(u32 *w, const T &a) { fp_freeze_words(w, a); }
    static ZKLC_M T unpack(const u32 *w) { return fp_from_words_raw(w); }
    static ZKLC_M void store(i32 *d, const T &a) {
#pragma unroll
        for (int k = 0; k < 10; k++) d[k] = a.v[k];
    }
    static ZKLC_M T load(const i32 *s) {
        T a;
#pragma unroll
        for (int k = 0; k < 10; k++) a.v[k] = s[k];
        return a;
    }
};
struct Fp2Field {
    typedef fp2 T;
    static constexpr int LIMBS = 20;
    static ZKLC_M T zero() { return fp2_zero(); }
    static ZKLC_M T one() { return fp2_one(); }
    static ZKLC_M T add(const T &a, const T &b) { return fp2_add(a, b); }
    static ZKLC_M T sub(const T &a, const T &b) { return fp2_sub(a, b); }
    static ZKLC_M T dbl(const T &a) { return fp2_dbl(a); }
    static ZKLC_M T neg(const T &a) { return fp2_neg(a); }
    static ZKLC_M T mul(const T &a, const T &b) { return fp2_mul(a, b); }
    static ZKLC_M T sqr(const T &a) { return fp2_sqr(a); }
    static ZKLC_M T inv(const T &a) { return fp2_inv(a); }
    static ZKLC_M T reduce(const T &a) { return fp2_reduce(a); }
    static ZKLC_M T select(const T &a, const T &b, u32 c) { return fp2_select(a, b, c); }
    static ZKLC_M u32 is_zero(const T &a) { return fp2_is_zero(a); }
    static ZKLC_M T from_gnark(const u32 *w) { return fp2_from_gnark(w); }
    static ZKLC_M void to_gnark(u32 *w, const T &a) { fp2_to_gnark(w, a); }
    static constexpr int PACKW = 16;
    static ZKLC_M T glv_beta() { return fp2_one(); }      // (no split on G2: never called)
    static ZKLC_M void pack(u32 *w, const T &a) {
        fp_freeze_words(w, a.c0);
        fp_freeze_words(w + 8, a.c1);
    }
    static ZKLC_M T unpack(const u32 *w) {
        T a;
        a.c0 = fp_from_words_raw(w);
        a.c1 = fp_from_words_raw(w + 8);
        return a;
    }
    static ZKLC_M void store(i32 *d, const T &a) {
        FpField::store(d, a.c0);
        FpField::store(d + 10, a.c1);
    }
    static ZKLC_M T load(const i32 *s) {
        T a;
        a.c0 = FpField::load(s);
        a.c1 = FpField::load(s + 10);
        return a;
    }
};

template <class F>
struct ec_xyzz {
    typename F::T X, Y, ZZ, ZZZ;
};

template <class F>
ZKLC_HD ec_xyzz<F> ec_infinity() {
    ec_xyzz<F> r;
    r.X = F::zero();
    r.Y = F::zero();
    r.ZZ = F::zero();
    r.ZZZ = F::zero();
    return r;
}
template <class F>
ZKLC_HD u32 ec_is_inf(const ec_xyzz<F> &p) {
    return F::is_zero(p.ZZ);
}

// dbl-2008-s-1
template <class F>
ZKLC_HD ec_xyzz<F> ec_double(const ec_xyzz<F> &p) {
    typedef typename F::T T;
    T U = F::dbl(p.Y);
    T V = F::sqr(U);
    T W = F::mul(U, V);
    T S = F::mul(p.X, V);
    T XX = F::sqr(p.X);
    T M = F::add(F::dbl(XX), XX);  // 3 X^2 (a = 0)
    ec_xyzz<F> r;
    r.X = F::sub(F::sqr(M), F::dbl(S));
    r.Y = F::sub(F::mul(M, F::sub(S, r.X)), F::mul(W, p.Y));
    r.ZZ = F::mul(V, p.ZZ);
    r.ZZZ = F::mul(W, p.ZZZ);
    return r;  // doubling infinity (ZZ = 0) gives ZZ = 0 again; these curves have no 2-torsion
}

// double of an affine point (mdbl-2008-s-1); x, y reduced
template <class F>
ZKLC_HD ec_xyzz<F> ec_double_affine(const typename F::T &x, const typename F::T &y) {
    typedef typename F::T T;
    T U = F::dbl(y);
    ec_xyzz<F> r;
    r.ZZ = F::sqr(U);
    r.ZZZ = F::mul(U, r.ZZ);
    T S = F::mul(x, r.ZZ);
    T XX = F::sqr(x);
    T M = F::add(F::dbl(XX), XX);
    r.X = F::sub(F::sqr(M), F::dbl(S));
    r.Y = F::sub(F::mul(M, F::sub(S, r.X)), F::mul(r.ZZZ, y));
    return r;
}

// the exceptional cases of p + (x2, y2): p at infinity (-> the point itself), P = Q (-> its double) and P = -Q (-> infinity); none
// of them needs p.  Kept out of line: the bucket loops of the MSM meet them once per bucket (the first point) or never, and inlined
// they doubled the loop's code.  The operands travel BY VALUE (registers): taken by reference they had to live in memory, and the
// compiler hoisted the stores of the copies above the rare branch -- six scratch stores (80 bytes) on EVERY addition of the slice
// kernel, 2.8 GB of write traffic per 2^22 multi-exponentiation (round 4, found in the WRITE_SIZE counter of the G2 kernel).
template <class F>
#if defined(__HIPCC__)
__device__ __attribute__((noinline))
#else
static
#endif
ec_xyzz<F> ec_add_affine_special(typename F::T x2, typename F::T y2, u32 p_inf, u32 same_y) {
    ec_xyzz<F> r;
    if (p_inf) {
        r.X = F::reduce(x2);
        r.Y = F::reduce(y2);
        r.ZZ = F::one();
        r.ZZZ = F::one();
    } else if (same_y) {
        r = ec_double_affine<F>(F::reduce(x2), F::reduce(y2));
    } else {
        r = ec_infinity<F>();
    }
    return r;
}

// the general case of p + (x2, y2) (madd-2008-s) and the flags of the exceptional ones: `special` = p at infinity or same x
// (P = +-Q), `same_y` = P = Q.  The result is meaningless when `special` is set.
template <class F>
ZKLC_HD ec_xyzz<F> ec_madd_core(const ec_xyzz<F> &p, const typename F::T &x2, const typename F::T &y2, u32 &p_inf, u32 &special, u32 &same_y) {
    typedef typename F::T T;
    T U2 = F::mul(x2, p.ZZ);
    T S2 = F::mul(y2, p.ZZZ);
    T Pp = F::sub(U2, p.X);
    T R = F::sub(S2, p.Y);
    p_inf = ec_is_inf(p);
    special = p_inf | F::is_zero(Pp);  // same x: P = +-Q
    T PP = F::sqr(Pp);
    T PPP = F::mul(Pp, PP);
    T Q = F::mul(p.X, PP);
    ec_xyzz<F> r;
    r.X = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    r.Y = F::sub(F::mul(R, F::sub(Q, r.X)), F::mul(p.Y, PPP));
    r.ZZ = F::mul(p.ZZ, PP);
    r.ZZZ = F::mul(p.ZZZ, PPP);
    same_y = 0;
    if (special) same_y = F::is_zero(R);
    return r;
}
// p + (x2, y2), (x2, y2) affine and finite (lazy from_gnark values allowed); neg = 1 adds (x2, -y2)
template <class F>
ZKLC_HD ec_xyzz<F> ec_add_affine(const ec_xyzz<F> &p, const typename F::T &x2, const typename F::T &y2in, u32 neg) {
    typedef typename F::T T;
    T y2 = F::select(y2in, F::neg(y2in), neg);
    u32 p_inf, special, same_y;
    ec_xyzz<F> r = ec_madd_core<F>(p, x2, y2, p_inf, special, same_y);
    if (special) r = ec_add_affine_special<F>(x2, y2, p_inf, same_y);
    return r;
}

// add-2008-s, general
template <class F>
ZKLC_HD ec_xyzz<F> ec_add(const ec_xyzz<F> &p, const ec_xyzz<F> &q) {
    typedef typename F::T T;
    u32 p_inf = ec_is_inf(p), q_inf = ec_is_inf(q);
    T U1 = F::mul(p.X, q.ZZ);
    T U2 = F::mul(q.X, p.ZZ);
    T S1 = F::mul(p.Y, q.ZZZ);
    T S2 = F::mul(q.Y, p.ZZZ);
    T Pp = F::sub(U2, U1);
    T R = F::sub(S2, S1);
    u32 same_x = F::is_zero(Pp);
    T PP = F::sqr(Pp);
    T PPP = F::mul(Pp, PP);
    T Q = F::mul(U1, PP);
    ec_xyzz<F> r;
    r.X = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    r.Y = F::sub(F::mul(R, F::sub(Q, r.X)), F::mul(S1, PPP));
    r.ZZ = F::mul(F::mul(p.ZZ, q.ZZ), PP);
    r.ZZZ = F::mul(F::mul(p.ZZZ, q.ZZZ), PPP);
    if (p_inf | q_inf | same_x) {
        if (p_inf) r = q;
        else if (q_inf) r = p;
        else if (F::is_zero(R)) r = ec_double(p);
        else r = ec_infinity<F>();
    }
    return r;
}

// affine (x, y) = (X / ZZ, Y / ZZZ) in gnark Montgomery words (2 * 8 * LIMBS/10 words); returns 1 for infinity (zeros)
template <class F>
ZKLC_HD u32 ec_to_affine_gnark(u32 *out, const ec_xyzz<F> &p) {
    typedef typename F::T T;
    const int W = 8 * F::LIMBS / 10;
    if (ec_is_inf(p)) {
        for (int i = 0; i < 2 * W; i++) out[i] = 0;
        return 1;
    }
    // one inversion: (ZZ * ZZZ)^-1 -> 1/ZZ = inv * ZZZ, 1/ZZZ = inv * ZZ
    T inv = F::inv(F::mul(p.ZZ, p.ZZZ));
    T x = F::mul(p.X, F::mul(inv, p.ZZZ));
    T y = F::mul(p.Y, F::mul(inv, p.ZZ));
    F::to_gnark(out, x);
    F::to_gnark(out + W, y);
    return 0;
}
