// Host-side (CPU) pieces of the plonky2 prover: the Fiat-Shamir transcript and PoseidonGate witness rows.
// These are sequential, tiny (a few hundred permutations per proof) and feed challenges to the kernels;
// they are compiled as plain C++ (csrc/plonky2_host.cpp) from the same poseidon_gl.cuh the kernels use.
#pragma once
#include <stdint.h>
#include <stddef.h>

void zklc_host_poseidon_permute(uint64_t *state12);
// hash_n_to_hash_no_pad (gnark-plonky2-verifier/poseidon/goldilocks.go:41-86)
void zklc_host_poseidon_hash_no_pad(const uint64_t *in, size_t n, uint64_t *out4);

// Duplex challenger (gnark-plonky2-verifier/challenger/challenger.go:42-166)
struct zklc_challenger {
    uint64_t state[12] = {};
    uint64_t in[8];
    int n_in = 0;
    uint64_t out[8];
    int n_out = 0;
    void observe(uint64_t e);
    void observe_many(const uint64_t *e, size_t n);
    // 32-byte digest: hasher 0 = 4 Goldilocks elements; hasher 1 = BN254 Fr little-endian, absorbed as
    // 7-byte limbs + the remaining bits (crypto/plonky2_bn128/src/config.rs:52-70 `to_vec`)
    void observe_hash(const uint8_t *h32, int hasher);
    uint64_t challenge();
    void duplex();
};

extern "C" int32_t zklc_poseidon_gl_gate_rows(const uint64_t *inputs, const uint64_t *swap, uint32_t n, uint64_t *rows);
