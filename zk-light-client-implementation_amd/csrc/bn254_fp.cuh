// BN254 base field Fp in Montgomery form on ten signed 26-bit limbs (R' = 2^260).
//
// Replaces gnark-crypto's `fp.Element` (ecc/bn254/fp, 4 x u64 Montgomery with
// R = 2^256; un-vendored: gnark-plonky2-verifier/go.mod:8-9) underneath
// `bn254.G1Affine.MultiExp`, reached from `groth16.Prove` at
// gnark-plonky2-verifier/cmd/web-api.go:77.  p as in
// contracts/hardhat/contracts/Verifier.sol:28-40.
//
// Same reasoning as fe25519.cuh: on gfx950 integer multiply-adds are full rate but
// carry flags are expensive, so products are accumulated column-wise in 64-bit
// registers (v_mad_i64_i32) with no carries, and the Montgomery reduction works on
// the 19 columns directly (one 26-bit quotient digit per column).
// Elements are LAZY: any signed limbs with |limb| < 2^30 and |value| < 16p are a
// legal operand of fp_mul / fp_freeze; fp_mul returns limbs |.| <= 2^25 (top limb small) and
// |value| < 1.5 p for operands up to 6p.  fp_freeze gives the canonical value.
// Boundary conversion: gnark's x * 2^256 is loaded as the integer 16 * (x * 2^256)
// = x * 2^260 (a lazy representative), and stored through one multiplication by
// 2^256 mod p (which divides by 16 in the Montgomery domain).
#pragma once
#include "common.cuh"

typedef int32_t i32;
typedef int64_t i64;

struct fp {
    i32 v[10];
};

#define FP_MASK26 0x3ffffff
#define FP_P26 {8191303, 2295222, 11064258, 38117831, 26706282, 6313495, 17062760, 41985761, 41083185, 792851}
#define FP_PINV26 8807305  // -p^-1 mod 2^26
// 1, 3 and 2^256 in the internal Montgomery domain (x * 2^260 mod p)
#define FP_ONE {{50128052, 8527933, 10126421, 19327654, 38373640, 6537298, 43123160, 29965846, 38673335, 509328}}
#define FP_THREE {{7975125, 23288579, 19315005, 19865131, 21305774, 13298400, 45197856, 47911778, 7827956, 735134}}
#define FP_R2 {{23522052, 36308806, 93062, 25580550, 10020373, 47440483, 22222336, 40216319, 27462970, 172314}}
#define FP_2P256 {{26152349, 55632753, 11787573, 10737436, 686315, 35541387, 48903927, 58506649, 63019527, 230045}}

ZKLC_HD fp fp_zero() {
    fp r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = 0;
    return r;
}
ZKLC_HD fp fp_add(const fp &a, const fp &b) {
    fp r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
ZKLC_HD fp fp_sub(const fp &a, const fp &b) {
    fp r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] - b.v[i];
    return r;
}
ZKLC_HD fp fp_neg(const fp &a) {
    fp r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = -a.v[i];
    return r;
}
ZKLC_HD fp fp_dbl(const fp &a) { return fp_add(a, a); }
ZKLC_HD fp fp_select(const fp &a, const fp &b, u32 cond) {
    fp r;
    i32 m = -(i32)cond;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
}

// Montgomery reduction of 19 signed columns (value = sum t[k] 2^(26k)) -> value / 2^260 mod p
ZKLC_HD fp fp_mont_reduce(i64 *t) {
    const i32 P[10] = FP_P26;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        i32 m = (i32)(((u32)t[i] * (u32)FP_PINV26) & FP_MASK26);  // quotient digit in [0, 2^26)
#pragma unroll
        for (int j = 0; j < 10; j++) t[i + j] += (i64)m * P[j];
        t[i + 1] += t[i] >> 26;  // exact: t[i] is now a multiple of 2^26
    }
    // carry-normalise t[10..19] into ten limbs (centred, top limb takes the rest)
    fp r;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        i64 c = (t[10 + j] + ((i64)1 << 25)) >> 26;
        t[11 + j] += c;
        r.v[j] = (i32)(t[10 + j] - (c << 26));
    }
    r.v[9] = (i32)t[19];
    return r;
}

ZKLC_HD fp fp_mul(const fp &a, const fp &b) {
    i64 t[20];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        i64 acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            int j = k - i;
            if (j >= 0 && j < 10) acc += (i64)a.v[i] * b.v[j];
        }
        t[k] = acc;
    }
    t[19] = 0;
    return fp_mont_reduce(t);
}

ZKLC_HD fp fp_sqr(const fp &a) {
    i32 a2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) a2[i] = 2 * a.v[i];
    i64 t[20];
#pragma unroll
    for (int k = 0; k < 19; k++) {
        i64 acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            int j = k - i;
            if (j < i || j >= 10) continue;
            acc += (i64)(i == j ? a.v[i] : a2[i]) * a.v[j];
        }
        t[k] = acc;
    }
    t[19] = 0;
    return fp_mont_reduce(t);
}

// canonical value in [0, p) as 8 little-endian u32 words (still in the internal Montgomery domain)
ZKLC_HD void fp_freeze_words(u32 *out, const fp &a) {
    const i32 P[10] = FP_P26;
    // v + 16p is positive for every legal lazy element (|v| < 16p); propagate to limbs in [0, 2^26)
    i64 l[10];
    i64 c = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        c += (i64)a.v[i] + 16 * (i64)P[i];
        l[i] = c & FP_MASK26;
        c >>= 26;
    }
    l[9] += c << 26;  // value < 32p < 2^259: the top limb absorbs what is left
    // subtract 16p, 8p, 4p, 2p, p while the value stays non-negative
#pragma unroll
    for (int k = 4; k >= 0; k--) {
        i64 d[10];
        i64 borrow = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            i64 x = l[i] - ((i64)P[i] << k) + borrow;
            if (i < 9) {
                d[i] = x & FP_MASK26;
                borrow = x >> 26;
            } else {
                d[i] = x;
            }
        }
        bool neg = d[9] < 0;
#pragma unroll
        for (int i = 0; i < 10; i++) l[i] = neg ? l[i] : d[i];
    }
    // pack 10 x 26 bits (value < p < 2^254) into 8 words
    u64 acc = 0;
    int bits = 0, w = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        acc |= (u64)l[i] << bits;
        bits += 26;
        if (bits >= 32) {
            out[w++] = (u32)acc;
            acc >>= 32;
            bits -= 32;
        }
    }
    if (w < 8) out[w] = (u32)acc;
}

// 8 little-endian words (any value < 2^256) -> limbs of that integer
ZKLC_HD fp fp_from_words_raw(const u32 *w) {
    fp r;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        int bit = 26 * i, wi = bit >> 5, sh = bit & 31;
        u64 x = (u64)w[wi] >> sh;
        if (wi + 1 < 8) x |= (u64)w[wi + 1] << (32 - sh);
        r.v[i] = (i32)(x & FP_MASK26);
    }
    return r;
}

// gnark-crypto Montgomery form (x * 2^256 mod p, 4 x u64 little-endian = 8 words) -> internal.
// The result is the LAZY integer 16 * (x * 2^256) (< 16p, limbs < 2^30): legal only as a direct
// operand of fp_mul / fp_sqr; pass it through fp_reduce before adding or freezing.
ZKLC_HD fp fp_from_gnark(const u32 *w) {
    fp r = fp_from_words_raw(w);
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] <<= 4;  // x * 2^260 as the lazy integer 16 * (x * 2^256)
    return r;
}
// internal -> gnark-crypto Montgomery form, canonical
ZKLC_HD void fp_to_gnark(u32 *out, const fp &a) {
    const fp c = FP_2P256;
    fp_freeze_words(out, fp_mul(a, c));
}

// same element, |value| < 1.2 p, limbs <= 2^25
ZKLC_HD fp fp_reduce(const fp &a) {
    const fp one = FP_ONE;
    return fp_mul(a, one);
}

ZKLC_HD u32 fp_is_zero(const fp &a) {
    u32 w[8];
    fp_freeze_words(w, a);
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= w[i];
    return o == 0;
}

// a^(p-2) (used once per MSM for the affine output, never in the bucket loops)
ZKLC_HD fp fp_inv(const fp &a) {
    // p - 2, little-endian 32-bit words
    const u32 E[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    const fp one = FP_ONE;
    fp r = one, x = a;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int i = 0; i < 254; i++) {
        if ((E[i >> 5] >> (i & 31)) & 1) r = fp_mul(r, x);
        x = fp_sqr(x);
    }
    return r;
}
