// Poseidon over BN254 Fr (iden3 parameters: t = 4, x^5, 8 full + 56 partial rounds) and the
// plonky2 hasher built on it -- the Merkle / Fiat-Shamir-cap hasher of the LAST recursion
// (PoseidonBN128GoldilocksConfig) so that the gnark wrap circuit is cheap.
// Follows crypto/plonky2_bn128/src/poseidon_bn128.rs:18-108 (permution, ark, exp5,
// full_rounds, partial_rounds, mix) and crypto/plonky2_bn128/src/config.rs:132-199
// (hash_no_pad packing: 3 Goldilocks elements per Fr as little-endian u64 limbs, 3 Fr per
// permutation into state[1..4], digest = state[0]; hash_or_noop; two_to_one).
// One lane = one state (4 x 10 VGPRs); constants via wave-uniform loads.
#pragma once
#include "bn254_fr.cuh"

#if defined(__HIPCC__)
#define ZKLC_CONST_ARRAY __device__ __constant__ const
#else
#define ZKLC_CONST_ARRAY static const
#endif
#include "poseidon_bn254_constants.inc"

ZKLC_HD fr pbn_const(const i32 *tab, int idx) {
    fr r;
#pragma unroll
    for (int k = 0; k < 10; k++) r.v[k] = tab[idx * 10 + k];
    return r;
}
ZKLC_HD fr pbn_exp5(const fr &x) {
    fr x2 = fr_sqr(x);
    return fr_mul(fr_sqr(x2), x);
}
// state <- state^T * M  (result_i = sum_j M[j][i] * state_j, poseidon_bn128.rs:92-108)
ZKLC_HD void pbn_mix(fr *s, const i32 *m) {
    fr o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        fr acc = fr_mul(pbn_const(m, 0 * 4 + i), s[0]);
#pragma unroll
        for (int j = 1; j < 4; j++) acc = fr_add(acc, fr_mul(pbn_const(m, j * 4 + i), s[j]));
        o[i] = fr_reduce(acc);  // sum of four products: bring the limbs back down before the next x^5
    }
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = o[i];
}

ZKLC_HD void poseidon_bn254_permute(fr *s) {
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = fr_add(s[i], pbn_const(PBN_C, i));
    // first half of the full rounds (3 with M, the 4th with P)
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = fr_add(pbn_exp5(s[i]), pbn_const(PBN_C, (r + 1) * 4 + i));
        pbn_mix(s, r < 3 ? PBN_M : PBN_P);
    }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 56; r++) {
        fr s0 = fr_add(pbn_exp5(s[0]), pbn_const(PBN_C, 20 + r));
        fr n0 = fr_mul(pbn_const(PBN_S, 7 * r), s0);
#pragma unroll
        for (int j = 1; j < 4; j++) n0 = fr_add(n0, fr_mul(pbn_const(PBN_S, 7 * r + j), s[j]));
#pragma unroll
        for (int k = 1; k < 4; k++) s[k] = fr_reduce(fr_add(s[k], fr_mul(s0, pbn_const(PBN_S, 7 * r + 4 + k - 1))));
        s[0] = fr_reduce(n0);
    }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = pbn_exp5(s[i]);
        if (r < 3) {
#pragma unroll
            for (int i = 0; i < 4; i++) s[i] = fr_add(s[i], pbn_const(PBN_C, 20 + 56 + r * 4 + i));
        }
        pbn_mix(s, PBN_M);
    }
}

// Fr (Montgomery) from up to three Goldilocks elements read at in[k * stride]: value = e0 + e1 2^64 + e2 2^128
ZKLC_HD fr pbn_pack3(const u64 *in, size_t stride, u32 count) {
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        u64 e = (u32)k < count ? in[(size_t)k * stride] : 0;
        w[2 * k] = (u32)e;
        w[2 * k + 1] = (u32)(e >> 32);
    }
    w[6] = w[7] = 0;
    return fr_from_regular(w);
}

// hash_or_noop (config.rs:174-186) / hash_no_pad (:139-171) of `len` Goldilocks elements in[i * stride];
// digest as 8 LE words of the REGULAR (non-Montgomery) canonical Fr value = PoseidonBN128HashOut::to_bytes.
ZKLC_HD void poseidon_bn254_hash_or_noop(const u64 *in, size_t stride, u32 len, u32 *out8) {
    if (len <= 3) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            u64 e = (u32)k < len ? in[(size_t)k * stride] : 0;
            out8[2 * k] = (u32)e;
            out8[2 * k + 1] = (u32)(e >> 32);
        }
        out8[6] = out8[7] = 0;
        return;
    }
    fr s[4];
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = fr_zero();
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (u32 off = 0; off < len; off += 9) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            u32 o = off + 3 * j;
            if (o < len) s[j + 1] = pbn_pack3(in + (size_t)o * stride, stride, len - o < 3 ? len - o : 3);
        }
        poseidon_bn254_permute(s);
    }
    fr_to_regular(out8, s[0]);
}

// two_to_one(l, r) = permute([0, 0, l, r])[0]; l, r, out = regular canonical words
ZKLC_HD void poseidon_bn254_two_to_one(const u32 *l8, const u32 *r8, u32 *out8) {
    fr s[4];
    s[0] = fr_zero();
    s[1] = fr_zero();
    s[2] = fr_from_regular(l8);
    s[3] = fr_from_regular(r8);
    poseidon_bn254_permute(s);
    fr_to_regular(out8, s[0]);
}
