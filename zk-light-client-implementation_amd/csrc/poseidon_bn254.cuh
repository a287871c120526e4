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
// state <- state^T * M  (result_i = sum_j M[j][i] * state_j, poseidon_bn128.rs:92-108): the four products of an output share
// ONE Montgomery reduction (fr_acc_*: 400 multiply-adds + one reduction instead of four multiplications and a fifth by one)
ZKLC_HD void pbn_mix(fr *s, const i32 *m) {
    fr o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        i64 t[20];
        fr_acc_zero(t);
#pragma unroll
        for (int j = 0; j < 4; j++) fr_acc_mul(t, pbn_const(m, j * 4 + i), s[j]);
        o[i] = fr_acc_reduce(t);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = o[i];
}

// one partial round (poseidon_bn128.rs:60-80, the sparse-matrix form): s0 <- s0^5 + c; new s0 = <S[0..4), (s0, s1, s2, s3)>;
// s_k += s0 * S[4 + k - 1] -- the dot product is one reduction, each update one (the old value enters as the product s_k * 1, so
// that the reduction keeps it below 1.2 r: added past the reduction it would grow by up to r per round)
ZKLC_HD void pbn_partial_round(fr *s, int r) {
    fr s0 = fr_add(pbn_exp5(s[0]), pbn_const(PBN_C, 20 + r));
    const fr one = FR_ONE;
    i64 t[20];
    fr_acc_zero(t);
    fr_acc_mul(t, pbn_const(PBN_S, 7 * r), s0);
#pragma unroll
    for (int j = 1; j < 4; j++) fr_acc_mul(t, pbn_const(PBN_S, 7 * r + j), s[j]);
    fr n0 = fr_acc_reduce(t);
#pragma unroll
    for (int k = 1; k < 4; k++) {
        fr_acc_zero(t);
        fr_acc_mul(t, s0, pbn_const(PBN_S, 7 * r + 4 + k - 1));
        fr_acc_mul(t, s[k], one);
        s[k] = fr_acc_reduce(t);
    }
    s[0] = n0;
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
    for (int r = 0; r < 56; r++) pbn_partial_round(s, r);
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

// ---- the permutation spread over FOUR lanes (one state word each), for the small trees of the last recursion
// A tree level with <= 2^14 nodes is a few hundred waves of single permutations: latency, not throughput.  Lane q of a quad holds
// state word q; a round is x^5 on the own word (every lane in the full rounds, lane 0 in the partial ones -- the others compute and
// discard), a quad broadcast of the four words (DPP quad_perm on the device) and ONE four-term dot product per lane: row q of the
// matrix in the full rounds; in a partial round lane 0 takes <S[0..4), v> and lane k >= 1 takes S[3 + k] v_0 + 1 v_k as a dot
// product with the coefficients (S[3 + k], .., 1 at k, ..).  ~1 500 instruction-times per round instead of ~2 400 (partial) /
// ~5 900 (full) on one lane.  These two functions are the per-lane step, shared with tests/hostsim (which walks the four lanes).
ZKLC_HD fr pbn_coop_mix_lane(u32 q, const fr *v, const i32 *m) {
    i64 t[20];
    fr_acc_zero(t);
#pragma unroll
    for (int j = 0; j < 4; j++) fr_acc_mul(t, pbn_const(m, j * 4 + (int)q), v[j]);
    return fr_acc_reduce(t);
}
ZKLC_HD fr pbn_coop_partial_lane(u32 q, const fr *v, int r) {
    const fr one = FR_ONE;
    i64 t[20];
    fr_acc_zero(t);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        fr c = pbn_const(PBN_S, q == 0 ? 7 * r + j : 7 * r + 3 + (int)q);      // lane k >= 1: S[3 + k] (used for j = 0)
        fr alt = fr_select(fr_zero(), one, (u32)j == q);                            // lane k >= 1, j >= 1: 1 at j = k, else 0
        c = fr_select(c, alt, (q != 0) & (j != 0));
        fr_acc_mul(t, c, v[j]);
    }
    return fr_acc_reduce(t);
}
// the own-word part of a round: state word q before the broadcast
ZKLC_HD fr pbn_coop_pre_lane(u32 q, const fr &s, int phase, int r) {
    if (phase == 0) return fr_add(pbn_exp5(s), pbn_const(PBN_C, (r + 1) * 4 + (int)q));           // first four full rounds
    if (phase == 1) return fr_select(s, fr_add(pbn_exp5(s), pbn_const(PBN_C, 20 + r)), q == 0);   // partial rounds: word 0 only
    fr e = pbn_exp5(s);                                                                            // last four full rounds
    return r < 3 ? fr_add(e, pbn_const(PBN_C, 20 + 56 + r * 4 + (int)q)) : e;
}
#if defined(__HIPCC__)
ZKLC_D void pbn_quad_gather(fr *v, const fr &s) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int k = 0; k < 10; k++) {
        v[0].v[k] = __builtin_amdgcn_mov_dpp(s.v[k], 0x00, 0xf, 0xf, false);   // quad_perm [0,0,0,0] .. [3,3,3,3]
        v[1].v[k] = __builtin_amdgcn_mov_dpp(s.v[k], 0x55, 0xf, 0xf, false);
        v[2].v[k] = __builtin_amdgcn_mov_dpp(s.v[k], 0xaa, 0xf, 0xf, false);
        v[3].v[k] = __builtin_amdgcn_mov_dpp(s.v[k], 0xff, 0xf, 0xf, false);
    }
#endif
}
// s = state word q of this quad's permutation (q = lane & 3; all four lanes of a quad must be active)
ZKLC_D void poseidon_bn254_permute_coop(fr &s, u32 q) {
    fr v[4];
    s = fr_add(s, pbn_const(PBN_C, (int)q));
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        s = pbn_coop_pre_lane(q, s, 0, r);
        pbn_quad_gather(v, s);
        s = pbn_coop_mix_lane(q, v, r < 3 ? PBN_M : PBN_P);
    }
#pragma unroll 1
    for (int r = 0; r < 56; r++) {
        s = pbn_coop_pre_lane(q, s, 1, r);
        pbn_quad_gather(v, s);
        s = pbn_coop_partial_lane(q, v, r);
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        s = pbn_coop_pre_lane(q, s, 2, r);
        pbn_quad_gather(v, s);
        s = pbn_coop_mix_lane(q, v, PBN_M);
    }
}
#endif
// the same walk over an explicit array of the four lanes (the CPU check of the per-lane functions)
ZKLC_HD void poseidon_bn254_permute_coop_ref(fr *s4) {
    fr n[4];
    for (u32 q = 0; q < 4; q++) s4[q] = fr_add(s4[q], pbn_const(PBN_C, (int)q));
    for (int phase = 0; phase < 3; phase++) {
        int rounds = phase == 1 ? 56 : 4;
        for (int r = 0; r < rounds; r++) {
            for (u32 q = 0; q < 4; q++) s4[q] = pbn_coop_pre_lane(q, s4[q], phase, r);
            for (u32 q = 0; q < 4; q++)
                n[q] = phase == 1 ? pbn_coop_partial_lane(q, s4, r) : pbn_coop_mix_lane(q, s4, phase == 0 && r == 3 ? PBN_P : PBN_M);
            for (u32 q = 0; q < 4; q++) s4[q] = n[q];
        }
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
