// One Ed25519 verification = one lane.  Semantics of the reference's native
// pre-check `sig.verify(msg, &pk)` at
// near_bft_finality/src/prove_block_data/signatures.rs:79 (ed25519-dalek
// non-strict verify): reject s >= l, reject an undecodable A, then
//     R' = [s]B + [h](-A),  h = SHA512(R || A || M) mod l,
// and accept iff compress(R') equals the 32 signature bytes of R.  On honest
// signatures this is the same predicate as the in-tree restatement
// crypto/plonky2_ed25519/src/curve/eddsa.rs:33-58 ([s]B == R + [h]A).
//
// Double-scalar multiplication: Straus with shared doublings,
//   [h](-A): signed radix-16 (64 digits in [-8,7]) over a per-signature table
//            of 8 cached multiples 1(-A) .. 8(-A),
//   [s]B   : signed radix-256 (32 digits in [-128,127]) over a constant table
//            of 128 affine-niels multiples 1B .. 128B shared by all lanes.
#pragma once
#include "ge25519.cuh"
#include "sc25519.cuh"
#include "sha512.cuh"

#define ZKLC_ED_BTABLE 128  // entries j = 1..128 of j*B

// Builds entry j (1-based) of the base table; run once at context creation.
ZKLC_HD ge_niels ed25519_base_table_entry(u32 j) {
    ge_p3 b = ge_base();
    ge_cached bc = ge_to_cached(b);
    ge_p3 acc = b;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (u32 i = 1; i < j; i++) acc = ge_add_cached<true>(acc, bc, 0);
    return ge_to_niels(acc);
}

// pk_w: 8 LE words, sig_w: 16 LE words (R || s), msg/msg_len.  `tab` = 8
// ge_cached of private storage for this lane.  Returns 1 (valid) or 0.
ZKLC_HD u32 ed25519_verify_one(const u32 *pk_w, const u32 *sig_w, const uint8_t *msg, u32 msg_len, const ge_niels *btab,
                               ge_cached *tab) {
    u32 ok = sc_is_canonical(sig_w + 8);

    // h = SHA512(R || A || M) mod l
    u32 prefix[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        prefix[i] = sig_w[i];
        prefix[8 + i] = pk_w[i];
    }
    u64 hs[8];
    sha512_hash_t<true>(prefix, msg, msg_len, hs);
    u32 hx[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        // digest bytes are the big-endian words; the scalar is the digest read little-endian
        u64 sw = sha_bswap64(hs[i]);
        hx[2 * i] = (u32)sw;
        hx[2 * i + 1] = (u32)(sw >> 32);
    }
    u32 h[8];
    sc_reduce512(h, hx);

    // -A and its multiples
    ge_p3 A;
    ok &= ge_decompress(A, pk_w);
    A.X = fe_neg(A.X);
    A.T = fe_neg(A.T);
    {
        ge_cached c1 = ge_to_cached(A);
        tab[0] = c1;
        ge_p3 acc = A;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (int j = 1; j < 8; j++) {
            acc = ge_add_cached<true>(acc, c1, 0);
            tab[j] = ge_to_cached(acc);
        }
    }

    // signed-window shift registers
    u32 hreg[8], sreg[8];
    sc_add_pattern(hreg, h, 0x88888888u);
    sc_add_pattern(sreg, sig_w + 8, 0x80808080u);

    ge_p3 acc = ge_identity();
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int i = 63; i >= 0; i--) {
        if (i != 63) {
            acc = ge_double<false>(acc);
            acc = ge_double<false>(acc);
            acc = ge_double<false>(acc);
            acc = ge_double<true>(acc);
        }
        {
            int dgt = (int)sc_shl_take(hreg, 4) - 8;  // [-8, 7]
            u32 neg = dgt < 0;
            u32 mag = neg ? (u32)(-dgt) : (u32)dgt;  // 0..8
            if (mag != 0) {
                ge_cached q = tab[mag - 1];
                acc = ge_add_cached<true>(acc, q, neg);
            }
        }
        if ((i & 1) == 0) {
            int dgt = (int)sc_shl_take(sreg, 8) - 128;  // [-128, 127]
            u32 neg = dgt < 0;
            u32 mag = neg ? (u32)(-dgt) : (u32)dgt;  // 0..128
            if (mag != 0) {
                ge_niels q = btab[mag - 1];
                acc = ge_add_niels<true>(acc, q, neg);
            }
        }
    }

    u32 rc[8];
    ge_compress(rc, acc);
    u32 diff = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) diff |= rc[i] ^ sig_w[i];
    return ok & (diff == 0);
}
