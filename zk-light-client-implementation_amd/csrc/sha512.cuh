// SHA-512 compression for the Ed25519 challenge hash h = SHA512(R || A || M).
//
// Restates the bit-level circuit of crypto/plonky2_sha512/src/circuit.rs:11-39
// (H, K), :141-224 (Sigma/sigma), :229-275 (ch/maj), :308-435 (schedule +
// rounds) as native 64-bit ALU code; native call site:
// crypto/plonky2_ed25519/src/curve/eddsa.rs:40-42.
#pragma once
#include "common.cuh"

#if defined(__HIPCC__)
__device__ __constant__
#else
static const
#endif
u64 SHA512_K[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
    0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
    0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
    0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
    0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
    0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
    0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
    0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
    0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
    0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
    0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
    0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
    0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};

ZKLC_HD u64 rotr64(u64 x, int n) { return (x >> n) | (x << (64 - n)); }

ZKLC_HD void sha512_init(u64 *h) {
    h[0] = 0x6a09e667f3bcc908ULL;
    h[1] = 0xbb67ae8584caa73bULL;
    h[2] = 0x3c6ef372fe94f82bULL;
    h[3] = 0xa54ff53a5f1d36f1ULL;
    h[4] = 0x510e527fade682d1ULL;
    h[5] = 0x9b05688c2b3e6c1fULL;
    h[6] = 0x1f83d9abfb41bd6bULL;
    h[7] = 0x5be0cd19137e2179ULL;
}

// one 1024-bit block; w[16] big-endian words, clobbered (rolling schedule)
ZKLC_HD void sha512_compress(u64 *h, u64 *w) {
    u64 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int t0 = 0; t0 < 80; t0 += 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (t0) {
                u64 w15 = w[(j + 1) & 15], w2 = w[(j + 14) & 15];
                u64 s0 = rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7);
                u64 s1 = rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6);
                w[j] = w[j] + s0 + w[(j + 9) & 15] + s1;
            }
            u64 S1 = rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41);
            u64 ch = (e & f) ^ (~e & g);
            u64 t1 = hh + S1 + ch + SHA512_K[t0 + j] + w[j];
            u64 S0 = rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39);
            u64 mj = (a & b) ^ (a & c) ^ (b & c);
            u64 t2 = S0 + mj;
            hh = g;
            g = f;
            f = e;
            e = d + t1;
            d = c;
            c = b;
            b = a;
            a = t1 + t2;
        }
    }
    h[0] += a;
    h[1] += b;
    h[2] += c;
    h[3] += d;
    h[4] += e;
    h[5] += f;
    h[6] += g;
    h[7] += hh;
}

ZKLC_HD u64 sha_bswap64(u64 v) {
    return ((v & 0xffULL) << 56) | ((v & 0xff00ULL) << 40) | ((v & 0xff0000ULL) << 24) | ((v & 0xff000000ULL) << 8) |
           ((v >> 8) & 0xff000000ULL) | ((v >> 24) & 0xff0000ULL) | ((v >> 40) & 0xff00ULL) | (v >> 56);
}

// SHA-512 of  prefix(64 bytes, given as 16 little-endian u32 words) || msg[0..msg_len)
// when PREFIX, else of msg alone.  Digest = h[0..8) as big-endian u64 words.
// The 64 prefix bytes are exactly schedule words w[0..8) of block 0, so they
// stay in registers; message bytes are fetched one by one (for the NEAR
// approval message they are wave-uniform addresses -> broadcast loads).
template <bool PREFIX>
ZKLC_HD void sha512_hash_t(const u32 *prefix_words, const uint8_t *msg, u32 msg_len, u64 *h) {
    sha512_init(h);
    const u32 plen = PREFIX ? 64u : 0u;
    u32 total = plen + msg_len;
    u32 nblocks = (total + 1 + 16 + 127) / 128;
    u64 w[16];
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (u32 blk = 0; blk < nblocks; blk++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (PREFIX && k < 8) {
                if (blk == 0) {
                    w[k] = sha_bswap64((u64)prefix_words[2 * k] | ((u64)prefix_words[2 * k + 1] << 32));
                    continue;
                }
            }
            u64 x = 0;
            u32 base = blk * 128 + k * 8;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                u32 i = base + q;
                u32 byte = 0;
                if (i < total)
                    byte = msg[i - plen];
                else if (i == total)
                    byte = 0x80;
                x = (x << 8) | byte;
            }
            w[k] = x;
        }
        if (blk == nblocks - 1) w[15] = (u64)total * 8;  // message bit length (< 2^35)
        sha512_compress(h, w);
    }
}

ZKLC_HD void sha512_hash_msg(const uint8_t *msg, u32 len, u64 *h) { sha512_hash_t<false>(nullptr, msg, len, h); }
