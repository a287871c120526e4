// BN254 scalar field Fr (r = 2188...5617) on the same ten-limb Montgomery machinery as Fp
// (mont26_impl.inc).  Used by the Poseidon-BN254 hasher of the final plonky2 recursion:
// crypto/plonky2_bn128/src/utils.rs (`Fr`, derived with the `ff` crate) and
// crypto/plonky2_bn128/src/poseidon_bn128.rs.  R as in contracts/hardhat/contracts/Verifier.sol:34.
#pragma once
#include "common.cuh"

typedef int32_t i32;
typedef int64_t i64;

struct fr {
    i32 v[10];
};

#define FR_P26 {1, 8217852, 50926654, 18999013, 19411944, 6313495, 17062760, 41985761, 41083185, 792851}
#define FR_PINV26 67108863  // -r^-1 mod 2^26
#define FR_ONE {{67108780, 47897935, 17128349, 14695580, 47118280, 6537307, 43123160, 29965846, 38673335, 509328}}
#define FR_2P256 {{67108859, 26019603, 13802185, 39222659, 37158006, 35541387, 48903927, 58506649, 63019527, 230045}}
// (2^260)^2 mod r: fr_mul(raw integer limbs, FR_R2) enters the Montgomery domain
#define FR_R2 {{23963381, 3019956, 13491427, 51938696, 54788022, 45903444, 17033154, 51316088, 42967330, 41044}}

#define MONT_T fr
#define MONT_FN(name) fr_##name
#define MONT_P26 FR_P26
#define MONT_PINV26 FR_PINV26
#define MONT_ONE FR_ONE
#define MONT_2P256 FR_2P256
#define MONT_PM2_WORDS {0xefffffffu, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}
#include "mont26_impl.inc"
#undef MONT_T
#undef MONT_FN
#undef MONT_P26
#undef MONT_PINV26
#undef MONT_ONE
#undef MONT_2P256
#undef MONT_PM2_WORDS

// regular (non-Montgomery) canonical integer < 2^256 as 8 LE words -> Montgomery domain (reduced)
ZKLC_HD fr fr_from_regular(const u32 *w) {
    const fr r2 = FR_R2;
    return fr_mul(fr_from_words_raw(w), r2);
}
// Montgomery domain -> regular canonical integer (8 LE words)
ZKLC_HD void fr_to_regular(u32 *out, const fr &a) {
    fr one_raw = fr_zero();
    one_raw.v[0] = 1;
    fr_freeze_words(out, fr_mul(a, one_raw));  // a / 2^260
}
