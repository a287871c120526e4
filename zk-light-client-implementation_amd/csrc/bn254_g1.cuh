// BN254 G1 (y^2 = x^3 + 3 over Fp): the generic extended-Jacobian formulas of bn254_ec.cuh instantiated on Fp.
// Replaces gnark-crypto's G1 bucket arithmetic (g1JacExtended; un-vendored, gnark-plonky2-verifier/go.mod:9) under
// `groth16.Prove` -> `MultiExp` (gnark-plonky2-verifier/cmd/web-api.go:77).
#pragma once
#include "bn254_ec.cuh"

typedef ec_xyzz<FpField> g1_xyzz;
struct g1_aff {  // internal Montgomery limbs (possibly lazy, see fp_from_gnark); inf = 1: point at infinity
    fp x, y;
    u32 inf;
};

ZKLC_HD g1_xyzz g1_infinity() { return ec_infinity<FpField>(); }
ZKLC_HD u32 g1_is_inf(const g1_xyzz &p) { return ec_is_inf(p); }
ZKLC_HD g1_xyzz g1_from_affine(const g1_aff &a) {
    g1_xyzz r;
    r.X = fp_reduce(a.x);
    r.Y = fp_reduce(a.y);
    r.ZZ = a.inf ? fp_zero() : FpField::one();
    r.ZZZ = r.ZZ;
    return r;
}
ZKLC_HD g1_xyzz g1_double(const g1_xyzz &p) { return ec_double(p); }
ZKLC_HD g1_xyzz g1_add_affine(const g1_xyzz &p, const fp &x2, const fp &y2, u32 neg) { return ec_add_affine<FpField>(p, x2, y2, neg); }
ZKLC_HD g1_xyzz g1_add(const g1_xyzz &p, const g1_xyzz &q) { return ec_add(p, q); }
ZKLC_HD u32 g1_to_affine_gnark(u32 *out16, const g1_xyzz &p) { return ec_to_affine_gnark(out16, p); }
