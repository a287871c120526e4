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

#define FP_P26 {8191303, 2295222, 11064258, 38117831, 26706282, 6313495, 17062760, 41985761, 41083185, 792851}
#define FP_PINV26 8807305  // -p^-1 mod 2^26
// 1, 3 and 2^256 in the internal Montgomery domain (x * 2^260 mod p)
#define FP_ONE {{50128052, 8527933, 10126421, 19327654, 38373640, 6537298, 43123160, 29965846, 38673335, 509328}}
#define FP_THREE {{7975125, 23288579, 19315005, 19865131, 21305774, 13298400, 45197856, 47911778, 7827956, 735134}}
#define FP_R2 {{23522052, 36308806, 93062, 25580550, 10020373, 47440483, 22222336, 40216319, 27462970, 172314}}
#define FP_2P256 {{26152349, 55632753, 11787573, 10737436, 686315, 35541387, 48903927, 58506649, 63019527, 230045}}

#define MONT_T fp
#define MONT_FN(name) fp_##name
#define MONT_P26 FP_P26
#define MONT_PINV26 FP_PINV26
#define MONT_ONE FP_ONE
#define MONT_2P256 FP_2P256
#define MONT_PM2_WORDS {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}
#include "mont26_impl.inc"
#undef MONT_T
#undef MONT_FN
#undef MONT_P26
#undef MONT_PINV26
#undef MONT_ONE
#undef MONT_2P256
#undef MONT_PM2_WORDS
