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
//   [h](-A): signed radix-8 (85 digits in [-4,3]) over a per-signature table of
//            4 cached multiples 1(-A)..4(-A) kept in LDS (40 words x 4 entries
//            per lane, laid out [entry][word][lane] so that every lane hits its
//            own bank whatever entry its digit selects: conflict-free),
//   [s]B   : signed radix-256 (32 digits in [-128,127]) over a constant table
//            of 128 affine-niels multiples 1B .. 128B shared by all lanes.
// One doubling per loop step; the additions are wave-uniform branches on the
// step counter (r % 3 == 0, r % 8 == 0), only the table INDEX varies per lane.
#pragma once
#include "ge25519.cuh"
#include "sc25519.cuh"
#include "sha512.cuh"

#define ZKLC_ED_BTABLE 128  // entries j = 1..128 of j*B
// window width of the variable-base part: signed radix 2^W, 2^(W-1) table entries.
//   W=3: 85 digits, 4 entries, 640 B of LDS per lane (4 waves per CU)
//   W=2: 127 digits, 2 entries, 320 B of LDS per lane (8 waves per CU)
#ifndef ZKLC_ED_AWIN
#define ZKLC_ED_AWIN 3
#endif
#define ZKLC_ED_ATAB_ENTRIES (1 << (ZKLC_ED_AWIN - 1))
#define ZKLC_ED_ATAB_WORDS (ZKLC_ED_ATAB_ENTRIES * 40)  // i32 words per lane
#define ZKLC_ED_ADIGITS ((253 + ZKLC_ED_AWIN) / ZKLC_ED_AWIN)  // 85 (W=3), 127 (W=2): digits*W in {255, 254}

// Builds entry j (1-based) of the base table; run once at context creation.
ZKLC_HD ge_niels ed25519_base_table_entry(u32 j) {
    ge_p3 b = ge_base();
    ge_cached bc = ge_to_cached(b);
    ge_p3 acc = b;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (u32 i = 1; i < j; i++) acc = ge_add_cached(acc, bc, 0, true);
    return ge_to_niels(acc);
}

// per-lane table of cached points: word w of entry e lives at tab[(e*40 + w) * STRIDE]
template <int STRIDE>
ZKLC_HD void atab_store(i32 *tab, int e, const ge_cached &c) {
#pragma unroll
    for (int w = 0; w < 10; w++) {
        tab[((e * 40) + w) * STRIDE] = c.YpX.v[w];
        tab[((e * 40) + 10 + w) * STRIDE] = c.YmX.v[w];
        tab[((e * 40) + 20 + w) * STRIDE] = c.Z.v[w];
        tab[((e * 40) + 30 + w) * STRIDE] = c.T2d.v[w];
    }
}
template <int STRIDE>
ZKLC_HD ge_cached atab_load(const i32 *tab, u32 e) {
    ge_cached c;
    const i32 *p = tab + (size_t)e * 40 * STRIDE;
#pragma unroll
    for (int w = 0; w < 10; w++) {
        c.YpX.v[w] = p[w * STRIDE];
        c.YmX.v[w] = p[(10 + w) * STRIDE];
        c.Z.v[w] = p[(20 + w) * STRIDE];
        c.T2d.v[w] = p[(30 + w) * STRIDE];
    }
    return c;
}

// pk_w: 8 LE words, sig_w: 16 LE words (R || s), msg/msg_len.  `tab` = this
// lane's slice of the cached-point table (ZKLC_ED_ATAB_WORDS words at STRIDE).
// Returns 1 (valid) or 0.
template <int STRIDE>
ZKLC_HD u32 ed25519_verify_one(const u32 *pk_w, const u32 *sig_w, const uint8_t *msg, u32 msg_len, const ge_niels *btab,
                               i32 *tab) {
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

    // -A and its multiples 1..4
    ge_p3 A;
    ok &= ge_decompress(A, pk_w);
    A.X = fe_neg(A.X);
    A.T = fe_neg(A.T);
    {
        ge_cached c1 = ge_to_cached(A);
        atab_store<STRIDE>(tab, 0, c1);
        ge_p3 a2 = ge_double(A, true);
        atab_store<STRIDE>(tab, 1, ge_to_cached(a2));
#if ZKLC_ED_AWIN == 3
        ge_p3 a3 = ge_add_cached(a2, c1, 0, true);
        atab_store<STRIDE>(tab, 2, ge_to_cached(a3));
        ge_p3 a4 = ge_double(a2, true);
        atab_store<STRIDE>(tab, 3, ge_to_cached(a4));
#endif
    }

    // signed-window shift registers: digit_i = field_i(x + pattern) - half
    //   h: ADIGITS radix-2^W digits (255 or 254 bits) -> pre-shift so the top W bits are the top digit
    //   s: 32 radix-256 digits
    u32 hreg[8], sreg[8];
    {
#if ZKLC_ED_AWIN == 3
        const u32 PAT[8] = {0x24924924u, 0x49249249u, 0x92492492u, 0x24924924u, 0x49249249u, 0x92492492u, 0x24924924u, 0x49249249u};
#else
        const u32 PAT[8] = {0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0x2aaaaaaau};
#endif
        u64 c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (u64)h[i] + PAT[i];
            hreg[i] = (u32)c;
            c >>= 32;
        }
        (void)sc_shl_take(hreg, 256 - ZKLC_ED_ADIGITS * ZKLC_ED_AWIN);
    }
    sc_add_pattern(sreg, sig_w + 8, 0x80808080u);

    ge_p3 acc = ge_identity();
    const int RTOP = (ZKLC_ED_ADIGITS - 1) * ZKLC_ED_AWIN;  // 252
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = RTOP; r >= 0; r--) {
        bool add_a = (r % ZKLC_ED_AWIN) == 0;
        bool add_b = (r & 7) == 0;
        if (r != RTOP) acc = ge_double(acc, add_a || add_b);
        if (add_a) {
            int dgt = (int)sc_shl_take(hreg, ZKLC_ED_AWIN) - (1 << (ZKLC_ED_AWIN - 1));  // [-4, 3] / [-2, 1]
            u32 neg = dgt < 0;
            u32 mag = neg ? (u32)(-dgt) : (u32)dgt;  // 0..4
            if (mag != 0) {
                ge_cached q = atab_load<STRIDE>(tab, mag - 1);
                acc = ge_add_cached(acc, q, neg, true);
            }
        }
        if (add_b) {
            int dgt = (int)sc_shl_take(sreg, 8) - 128;  // [-128, 127]
            u32 neg = dgt < 0;
            u32 mag = neg ? (u32)(-dgt) : (u32)dgt;  // 0..128
            if (mag != 0) {
                ge_niels q = btab[mag - 1];
                acc = ge_add_niels(acc, q, neg, true);
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
