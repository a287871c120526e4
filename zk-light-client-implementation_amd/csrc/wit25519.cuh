// Register-resident arithmetic modulo p = 2^255 - 19 on eight 32-bit words for the witness generators of the non-native
// gadgets (crypto/plonky2_ed25519/src/gadgets/nonnative.rs:447-705: the generators compute a * b = q p + r, x^-1 and the
// quotient of x x^-1 - 1; gadgets/curve.rs:327-370: point decompression).  The generic big-integer code of
// plonky2_witness_ops.h keeps its numbers in indexed arrays (scratch memory on the GPU) and inverts with a 255-step
// square-and-multiply; the critical path of the Ed25519 circuit's witness holds 318 inversions and 1 272 multiplications, so on
// the device these run here instead: fully unrolled word arithmetic, and the inversions / square roots on the radix-2^25.5
// field code the signature-verification kernel uses (fe25519.cuh: addition chains, 11 multiplications + 254 squarings).
// Results are the canonical ones, i.e. identical to the generic path (checked on the CPU in tests/test_hostsim_ed25519.py).
#pragma once
#include "fe25519.cuh"

// modulus words as the generators' parameters hold them (u64 each); also called by the host-side scheduler
#if defined(__HIPCC__)
__host__
#endif
ZKLC_HD bool w25519_is_p(const int64_t *ml) {
    if ((u64)ml[0] != 0xFFFFFFEDu || (u64)ml[7] != 0x7FFFFFFFu) return false;
#pragma unroll
    for (int i = 1; i < 7; i++)
        if ((u64)ml[i] != 0xFFFFFFFFu) return false;
    return true;
}

// w (< 2^256) -> w mod p, canonical
ZKLC_HD void w25519_reduce(u32 *w) {
    // fold bit 255: 2^255 = 19; then one conditional subtraction of p
    u64 c = 19ULL * (w[7] >> 31);
    w[7] &= 0x7FFFFFFFu;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += w[i];
        w[i] = (u32)c;
        c >>= 32;
    }
    // w < 2^255 + 19: subtract p iff w >= p, i.e. iff w + 19 >= 2^255
    u32 t[8];
    u64 s = 19;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        s += w[i];
        t[i] = (u32)s;
        s >>= 32;
    }
    u32 ge = t[7] >> 31;      // w + 19 >= 2^255
    t[7] &= 0x7FFFFFFFu;
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = ge ? t[i] : w[i];
}

// 256 x 256 -> 512 bits, schoolbook by columns
ZKLC_HD void w25519_mul_wide(const u32 *a, const u32 *b, u32 *prod) {
    u64 lo = 0;      // column accumulator: low 64 bits
    u32 hi = 0;      // and its overflow
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int j = k - i;
            if (j >= 0 && j < 8) mac96(lo, hi, a[i], b[j]);
        }
        prod[k] = (u32)lo;
        lo = (lo >> 32) | ((u64)hi << 32);
        hi = 0;
    }
    prod[15] = (u32)lo;
}

// prod (512 bits) = q p + r with 0 <= r < p; q has nine words (q < 2^257)
ZKLC_HD void w25519_divmod(const u32 *prod, u32 *q, u32 *r) {
    // prod = H 2^255 + L;  H 2^255 = H p + 19 H  ->  q = H, rest = 19 H + L  (< 2^262);  repeat once; then at most two
    // subtractions of p
    u32 H[9], L[8];
#pragma unroll
    for (int i = 0; i < 8; i++) L[i] = prod[i];
    L[7] &= 0x7FFFFFFFu;
#pragma unroll
    for (int i = 0; i < 8; i++) H[i] = (prod[7 + i] >> 31) | (prod[8 + i] << 1);
    H[8] = prod[15] >> 31;
    // rest = 19 H + L: nine words
    u32 R[9];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += 19ULL * H[i] + L[i];
        R[i] = (u32)c;
        c >>= 32;
    }
    c += 19ULL * H[8];
    R[8] = (u32)c;
    // second fold: H2 = rest >> 255 (< 2^8), rest = 19 H2 + (rest mod 2^255)
    u32 H2 = (R[7] >> 31) | (R[8] << 1);
    R[7] &= 0x7FFFFFFFu;
    c = 19ULL * H2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += R[i];
        R[i] = (u32)c;
        c >>= 32;
    }
    // R < 2^255 + 2^13: one conditional subtraction (R >= p iff R + 19 >= 2^255; the sum may reach bit 255 itself)
    u32 extra = 0;
#pragma unroll 1
    for (int rep = 0; rep < 2; rep++) {
        u32 t[8];
        u64 s = 19;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            s += R[i];
            t[i] = (u32)s;
            s >>= 32;
        }
        u32 ge = (t[7] >> 31) | (u32)s;
        t[7] &= 0x7FFFFFFFu;     // - 2^255
#pragma unroll
        for (int i = 0; i < 8; i++) R[i] = ge ? t[i] : R[i];
        extra += ge;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = R[i];
    c = (u64)H2 + extra;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        c += H[i];
        q[i] = (u32)c;
        c >>= 32;
    }
}

// x^-1 mod p for canonical x != 0 (x = 0 gives 0)
ZKLC_HD void w25519_inv(const u32 *x, u32 *out) {
    fe_freeze_words(out, fe_invert(fe_from_words(x)));
}

// curve25519 point decompression as CurvePointDecompressionGenerator (gadgets/curve.rs:327-370) runs it: y = the low 255 bits,
// x = sqrt((y^2 - 1) / (d y^2 + 1)) with the requested parity; false when y is not the ordinate of a curve point.
ZKLC_HD bool w25519_decompress(const u32 *yw /* bit 255 ignored */, u32 sign, u32 *xw) {
    const fe D = FE_D, SQRTM1 = FE_SQRTM1;
    fe y = fe_from_words(yw);
    fe yy = fe_sqr(y);
    fe u = fe_sub(yy, fe_one());
    fe v = fe_add(fe_mul(D, yy), fe_one());
    fe xx = fe_mul(u, fe_invert(v));
    fe x = fe_mul(xx, fe_pow22523(xx));          // xx^((p + 3) / 8)
    if (!fe_eq(fe_sqr(x), xx)) x = fe_mul(x, SQRTM1);
    if (!fe_eq(fe_sqr(x), xx)) return false;
    if (fe_is_negative(x) != sign) x = fe_neg(x);
    fe_freeze_words(xw, x);
    return true;
}
