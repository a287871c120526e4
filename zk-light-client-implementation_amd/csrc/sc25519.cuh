// Scalars mod l = 2^252 + 27742317777372353535851937790883648493.
//
// Replaces Ed25519Scalar (crypto/plonky2_ed25519/src/field/ed25519_scalar.rs:17,96-101)
// and the BigUint `mod_floor` of curve/eddsa.rs:43-45 with a fixed Barrett
// reduction (HAC 14.42, b = 2^32, k = 8) on 32-bit limbs.
#pragma once
#include "common.cuh"

#define SC_L_INIT {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0x00000000u, 0x00000000u, 0x00000000u, 0x10000000u}
// mu = floor(2^512 / l), 9 limbs
#define SC_MU_INIT {0x0a2c131bu, 0xed9ce5a3u, 0x086329a7u, 0x2106215du, 0xffffffebu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x0000000fu}

// 1 if a (8 limbs) < l
ZKLC_HD u32 sc_is_canonical(const u32 *a) {
    const u32 Lc[8] = SC_L_INIT;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a[i] - (int64_t)Lc[i];
        c >>= 32;
    }
    return (u32)(c & 1);  // borrow out <=> a < l
}

// out[8] = x[16] mod l   (x = 512-bit little-endian)
ZKLC_HD void sc_reduce512(u32 *out, const u32 *x) {
    const u32 Lc[8] = SC_L_INIT;
    const u32 MU[9] = SC_MU_INIT;
    // q2 = (x >> 224) * mu ; we only need limbs 9..17 of the 18-limb product,
    // but the low columns feed carries, so run all columns.
    u32 q3[9];
    {
        u64 lo = 0;
        u32 hi = 0;
#pragma unroll
        for (int k = 0; k < 17; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                int j = k - i;
                if (j >= 0 && j < 9) mac96(lo, hi, x[7 + i], MU[j]);
            }
            if (k >= 9) q3[k - 9] = (u32)lo;
            lo = (lo >> 32) | ((u64)hi << 32);
            hi = 0;
        }
        q3[8] = (u32)lo;
    }
    // r2 = (q3 * l) mod 2^288
    u32 r2[9];
    {
        u64 lo = 0;
        u32 hi = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                int j = k - i;
                if (j >= 0 && j < 8) mac96(lo, hi, q3[i], Lc[j]);
            }
            r2[k] = (u32)lo;
            lo = (lo >> 32) | ((u64)hi << 32);
            hi = 0;
        }
    }
    // r = (x mod 2^288) - r2 (mod 2^288), then at most two subtractions of l
    u32 r[9];
    {
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            c += (int64_t)x[i] - (int64_t)r2[i];
            r[i] = (u32)c;
            c >>= 32;
        }
    }
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        u32 t[9];
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            c += (int64_t)r[i] - (int64_t)(i < 8 ? Lc[i] : 0u);
            t[i] = (u32)c;
            c >>= 32;
        }
        u32 keep = (u32)(c & 1);  // borrow -> r < l -> keep r
        u32 m = 0u - keep;
#pragma unroll
        for (int i = 0; i < 9; i++) r[i] = (r[i] & m) | (t[i] & ~m);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = r[i];
}

// out = a + c*(0x0101..01) pattern add used by the signed-window recodings:
//   radix-16  digits d_i = nibble_i(a + 0x88..8)  - 8   in [-8, 7]
//   radix-256 digits d_j = byte_j  (a + 0x80..80) - 128 in [-128, 127]
// valid for a < 2^253 (no overflow out of 256 bits).
ZKLC_HD void sc_add_pattern(u32 *out, const u32 *a, u32 pattern) {
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a[i] + pattern;
        out[i] = (u32)c;
        c >>= 32;
    }
}

// shift a 256-bit register left by `n` bits (0 < n < 32) and return the bits shifted out
ZKLC_HD u32 sc_shl_take(u32 *a, int n) {
    u32 out = a[7] >> (32 - n);
#pragma unroll
    for (int i = 7; i > 0; i--) a[i] = (a[i] << n) | (a[i - 1] >> (32 - n));
    a[0] <<= n;
    return out;
}
