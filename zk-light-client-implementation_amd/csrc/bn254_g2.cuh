// BN254 G2 (the sextic twist y^2 = x^3 + 3/(9 + u) over Fp2): bn254_ec.cuh instantiated on Fp2.
// Replaces gnark-crypto's G2 bucket arithmetic (g2JacExtended, un-vendored) under `groth16.Prove` -> `G2Affine.MultiExp`
// (the B2 term of the proof; gnark-plonky2-verifier/cmd/web-api.go:77).  Affine point at the ABI = X.A0, X.A1, Y.A0, Y.A1,
// each 4 little-endian u64 in Montgomery form (gnark-crypto's G2Affine memory layout); generator as in
// contracts/hardhat/contracts/Verifier.sol (the G2 constants of the pairing precompile input).
#pragma once
#include "bn254_ec.cuh"

typedef ec_xyzz<Fp2Field> g2_xyzz;
ZKLC_HD g2_xyzz g2_infinity() { return ec_infinity<Fp2Field>(); }
ZKLC_HD g2_xyzz g2_double(const g2_xyzz &p) { return ec_double(p); }
ZKLC_HD g2_xyzz g2_add_affine(const g2_xyzz &p, const fp2 &x2, const fp2 &y2, u32 neg) { return ec_add_affine<Fp2Field>(p, x2, y2, neg); }
ZKLC_HD g2_xyzz g2_add(const g2_xyzz &p, const g2_xyzz &q) { return ec_add(p, q); }
ZKLC_HD u32 g2_to_affine_gnark(u32 *out32, const g2_xyzz &p) { return ec_to_affine_gnark(out32, p); }
