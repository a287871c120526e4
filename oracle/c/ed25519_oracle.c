/* Ed25519 batch verification -- plain C restatement.  TEST INFRASTRUCTURE / CPU
 * BASELINE ONLY: used by tests (big-N parity), __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Never linked into libzklc_mi355.so.
 *
 * Follows the reference's native pre-check
 *   near_bft_finality/src/prove_block_data/signatures.rs:72-86 (`sig.verify`, :79)
 * with ed25519-dalek non-strict semantics (un-vendored dependency, Cargo.toml:20-22),
 * and agrees with the in-tree restatement crypto/plonky2_ed25519/src/curve/eddsa.rs:33-58
 * on honest signatures.  Pinned by oracle/ed25519_ref.py (itself pinned by the
 * reference's mainnet fixtures) in tests/test_oracle_c.py.
 *
 * Independent of the GPU code on purpose: radix-2^51 limbs with unsigned __int128
 * products, Barrett-free scalar reduction by repeated folding, 4-bit fixed windows.
 */
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint8_t u8;

/* ---------------- SHA-512 (crypto/plonky2_sha512/src/circuit.rs:11-39,308-435) ---------------- */
static const u64 K[80] = {
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
#define ROR(x, n) (((x) >> (n)) | ((x) << (64 - (n))))

typedef struct {
    u64 h[8];
    u8 buf[128];
    u64 len;
} sha512_ctx;

static void sha512_block(u64 *h, const u8 *p) {
    u64 w[80];
    for (int i = 0; i < 16; i++) {
        u64 x = 0;
        for (int j = 0; j < 8; j++) x = (x << 8) | p[8 * i + j];
        w[i] = x;
    }
    for (int i = 16; i < 80; i++) {
        u64 s0 = ROR(w[i - 15], 1) ^ ROR(w[i - 15], 8) ^ (w[i - 15] >> 7);
        u64 s1 = ROR(w[i - 2], 19) ^ ROR(w[i - 2], 61) ^ (w[i - 2] >> 6);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    u64 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 80; i++) {
        u64 t1 = hh + (ROR(e, 14) ^ ROR(e, 18) ^ ROR(e, 41)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        u64 t2 = (ROR(a, 28) ^ ROR(a, 34) ^ ROR(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
static void sha512_init(sha512_ctx *c) {
    static const u64 iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                              0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    memcpy(c->h, iv, 64);
    c->len = 0;
}
static void sha512_update(sha512_ctx *c, const u8 *p, u64 n) {
    while (n) {
        u64 off = c->len & 127, take = 128 - off;
        if (take > n) take = n;
        memcpy(c->buf + off, p, take);
        c->len += take; p += take; n -= take;
        if ((c->len & 127) == 0) sha512_block(c->h, c->buf);
    }
}
static void sha512_final(sha512_ctx *c, u8 *out) {
    u64 bits = c->len * 8, off = c->len & 127;
    c->buf[off++] = 0x80;
    if (off > 112) { memset(c->buf + off, 0, 128 - off); sha512_block(c->h, c->buf); off = 0; }
    memset(c->buf + off, 0, 128 - off);
    for (int i = 0; i < 8; i++) c->buf[127 - i] = (u8)(bits >> (8 * i));
    sha512_block(c->h, c->buf);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (u8)(c->h[i] >> (56 - 8 * j));
}
void zklc_oracle_sha512(const u8 *msg, u64 len, u8 *out) {
    sha512_ctx c;
    sha512_init(&c);
    sha512_update(&c, msg, len);
    sha512_final(&c, out);
}

/* ---------------- GF(2^255-19), radix 2^51 (field/ed25519_base.rs:99-116) ---------------- */
typedef struct { u64 v[5]; } fe;
#define M51 ((1ULL << 51) - 1)
static const fe FE_D = {{0x34dca135978a3ULL, 0x1a8283b156ebdULL, 0x5e7a26001c029ULL, 0x739c663a03cbbULL, 0x52036cee2b6ffULL}};
static const fe FE_SQRTM1 = {{0x61b274a0ea0b0ULL, 0xd5a5fc8f189dULL, 0x7ef5e9cbd0c60ULL, 0x78595a6804c9eULL, 0x2b8324804fc1dULL}};

static void fe_carry(fe *r) {
    u64 c;
    c = r->v[0] >> 51; r->v[0] &= M51; r->v[1] += c;
    c = r->v[1] >> 51; r->v[1] &= M51; r->v[2] += c;
    c = r->v[2] >> 51; r->v[2] &= M51; r->v[3] += c;
    c = r->v[3] >> 51; r->v[3] &= M51; r->v[4] += c;
    c = r->v[4] >> 51; r->v[4] &= M51; r->v[0] += 19 * c;
    c = r->v[0] >> 51; r->v[0] &= M51; r->v[1] += c;
}
static fe fe_add(fe a, fe b) { fe r; for (int i = 0; i < 5; i++) r.v[i] = a.v[i] + b.v[i]; fe_carry(&r); return r; }
static fe fe_sub(fe a, fe b) {
    /* a + 4p - b keeps limbs non-negative */
    fe r;
    r.v[0] = a.v[0] + 0x1fffffffffffb4ULL - b.v[0];
    for (int i = 1; i < 5; i++) r.v[i] = a.v[i] + 0x1ffffffffffffcULL - b.v[i];
    fe_carry(&r);
    return r;
}
static fe fe_mul(fe a, fe b) {
    u128 t[5];
    u64 b1 = b.v[1] * 19, b2 = b.v[2] * 19, b3 = b.v[3] * 19, b4 = b.v[4] * 19;
    t[0] = (u128)a.v[0] * b.v[0] + (u128)a.v[1] * b4 + (u128)a.v[2] * b3 + (u128)a.v[3] * b2 + (u128)a.v[4] * b1;
    t[1] = (u128)a.v[0] * b.v[1] + (u128)a.v[1] * b.v[0] + (u128)a.v[2] * b4 + (u128)a.v[3] * b3 + (u128)a.v[4] * b2;
    t[2] = (u128)a.v[0] * b.v[2] + (u128)a.v[1] * b.v[1] + (u128)a.v[2] * b.v[0] + (u128)a.v[3] * b4 + (u128)a.v[4] * b3;
    t[3] = (u128)a.v[0] * b.v[3] + (u128)a.v[1] * b.v[2] + (u128)a.v[2] * b.v[1] + (u128)a.v[3] * b.v[0] + (u128)a.v[4] * b4;
    t[4] = (u128)a.v[0] * b.v[4] + (u128)a.v[1] * b.v[3] + (u128)a.v[2] * b.v[2] + (u128)a.v[3] * b.v[1] + (u128)a.v[4] * b.v[0];
    fe r;
    u64 c;
    r.v[0] = (u64)t[0] & M51; c = (u64)(t[0] >> 51); t[1] += c;
    r.v[1] = (u64)t[1] & M51; c = (u64)(t[1] >> 51); t[2] += c;
    r.v[2] = (u64)t[2] & M51; c = (u64)(t[2] >> 51); t[3] += c;
    r.v[3] = (u64)t[3] & M51; c = (u64)(t[3] >> 51); t[4] += c;
    r.v[4] = (u64)t[4] & M51; c = (u64)(t[4] >> 51);
    r.v[0] += c * 19; c = r.v[0] >> 51; r.v[0] &= M51; r.v[1] += c;
    return r;
}
static fe fe_sqr(fe a) { return fe_mul(a, a); }
static fe fe_sqrn(fe a, int n) { while (n--) a = fe_sqr(a); return a; }
static void fe_chain(fe z, fe *z250, fe *z11) {
    fe z2 = fe_sqr(z), z9 = fe_mul(z, fe_sqrn(z2, 2));
    *z11 = fe_mul(z2, z9);
    fe z5 = fe_mul(z9, fe_sqr(*z11));
    fe z10 = fe_mul(fe_sqrn(z5, 5), z5), z20 = fe_mul(fe_sqrn(z10, 10), z10), z40 = fe_mul(fe_sqrn(z20, 20), z20);
    fe z50 = fe_mul(fe_sqrn(z40, 10), z10), z100 = fe_mul(fe_sqrn(z50, 50), z50), z200 = fe_mul(fe_sqrn(z100, 100), z100);
    *z250 = fe_mul(fe_sqrn(z200, 50), z50);
}
static fe fe_inv(fe z) { fe t, z11; fe_chain(z, &t, &z11); return fe_mul(fe_sqrn(t, 5), z11); }
static fe fe_pow22523(fe z) { fe t, z11; fe_chain(z, &t, &z11); return fe_mul(fe_sqrn(t, 2), z); }
static void fe_tobytes(u8 *s, fe a) {
    fe_carry(&a); fe_carry(&a);
    /* canonical: add 19, check bit 255, subtract accordingly */
    u64 q = (a.v[0] + 19) >> 51;
    q = (a.v[1] + q) >> 51; q = (a.v[2] + q) >> 51; q = (a.v[3] + q) >> 51; q = (a.v[4] + q) >> 51;
    a.v[0] += 19 * q;
    u64 c;
    c = a.v[0] >> 51; a.v[0] &= M51; a.v[1] += c;
    c = a.v[1] >> 51; a.v[1] &= M51; a.v[2] += c;
    c = a.v[2] >> 51; a.v[2] &= M51; a.v[3] += c;
    c = a.v[3] >> 51; a.v[3] &= M51; a.v[4] += c;
    a.v[4] &= M51;
    u64 w[4] = {a.v[0] | (a.v[1] << 51), (a.v[1] >> 13) | (a.v[2] << 38), (a.v[2] >> 26) | (a.v[3] << 25),
                (a.v[3] >> 39) | (a.v[4] << 12)};
    memcpy(s, w, 32);
}
static fe fe_frombytes(const u8 *s) { /* top bit ignored, value NOT checked for canonicity (dalek) */
    u64 w[4];
    memcpy(w, s, 32);
    fe r;
    r.v[0] = w[0] & M51;
    r.v[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
    r.v[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
    r.v[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
    r.v[4] = (w[3] >> 12) & M51;
    return r;
}
static int fe_iszero(fe a) { u8 s[32]; fe_tobytes(s, a); u8 o = 0; for (int i = 0; i < 32; i++) o |= s[i]; return o == 0; }
static int fe_isneg(fe a) { u8 s[32]; fe_tobytes(s, a); return s[0] & 1; }
static const fe FE_ZERO = {{0, 0, 0, 0, 0}}, FE_ONE = {{1, 0, 0, 0, 0}};
static fe fe_neg(fe a) { return fe_sub(FE_ZERO, a); }

/* ---------------- group (curve/curve_adds.rs:8-60, curve_types.rs:171-223) ---------------- */
typedef struct { fe X, Y, Z, T; } ge;
static ge ge_add(ge p, ge q) {
    fe d2 = fe_add(FE_D, FE_D);
    fe A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X)), B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    fe C = fe_mul(fe_mul(p.T, q.T), d2), D = fe_mul(p.Z, q.Z);
    D = fe_add(D, D);
    fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    ge r = {fe_mul(E, F), fe_mul(G, H), fe_mul(F, G), fe_mul(E, H)};
    return r;
}
static ge ge_dbl(ge p) {
    fe A = fe_sqr(p.X), B = fe_sqr(p.Y), C = fe_sqr(p.Z);
    C = fe_add(C, C);
    fe E = fe_sub(fe_sub(fe_sqr(fe_add(p.X, p.Y)), A), B), G = fe_sub(B, A), F = fe_sub(G, C), H = fe_sub(fe_neg(A), B);
    ge r = {fe_mul(E, F), fe_mul(G, H), fe_mul(F, G), fe_mul(E, H)};
    return r;
}
static const ge GE_ID = {{{0, 0, 0, 0, 0}}, {{1, 0, 0, 0, 0}}, {{1, 0, 0, 0, 0}}, {{0, 0, 0, 0, 0}}};
static int ge_decompress(ge *r, const u8 *s) {
    fe y = fe_frombytes(s), yy = fe_sqr(y);
    fe u = fe_sub(yy, FE_ONE), v = fe_add(fe_mul(yy, FE_D), FE_ONE);
    fe v3 = fe_mul(fe_sqr(v), v), v7 = fe_mul(fe_sqr(v3), v);
    fe x = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe chk = fe_mul(v, fe_sqr(x));
    int ok = 1;
    if (fe_iszero(fe_sub(chk, u))) { }
    else if (fe_iszero(fe_add(chk, u))) x = fe_mul(x, FE_SQRTM1);
    else ok = 0;
    if (fe_isneg(x) != (s[31] >> 7)) x = fe_neg(x);
    r->X = x; r->Y = y; r->Z = FE_ONE; r->T = fe_mul(x, y);
    return ok;
}
static void ge_compress(u8 *s, ge p) {
    fe zi = fe_inv(p.Z), x = fe_mul(p.X, zi), y = fe_mul(p.Y, zi);
    fe_tobytes(s, y);
    s[31] |= (u8)(fe_isneg(x) << 7);
}

/* ---------------- scalars (field/ed25519_scalar.rs:96-101) ---------------- */
/* l as 32 little-endian bytes */
static const u8 L_BYTES[32] = {0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                               0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10};
static int sc_lt_l(const u8 *s) {
    for (int i = 31; i >= 0; i--) {
        if (s[i] < L_BYTES[i]) return 1;
        if (s[i] > L_BYTES[i]) return 0;
    }
    return 0;
}
/* x (64 bytes LE) mod l, bit-serial shift-subtract (slow but obviously right) */
static void sc_reduce64(u8 *out, const u8 *x) {
    u64 r[5] = {0, 0, 0, 0, 0}; /* 320-bit accumulator, value < 2l */
    u64 l[5] = {0, 0, 0, 0, 0};
    memcpy(l, L_BYTES, 32);
    for (int bit = 511; bit >= 0; bit--) {
        /* r = 2r + bit */
        for (int i = 4; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
        r[0] = (r[0] << 1) | ((x[bit >> 3] >> (bit & 7)) & 1);
        /* if r >= l: r -= l */
        int ge = 1;
        for (int i = 4; i >= 0; i--) {
            if (r[i] > l[i]) break;
            if (r[i] < l[i]) { ge = 0; break; }
        }
        if (ge) {
            u64 borrow = 0;
            for (int i = 0; i < 5; i++) {
                u128 d = (u128)r[i] - l[i] - borrow;
                r[i] = (u64)d;
                borrow = (u64)(d >> 64) & 1;
            }
        }
    }
    memcpy(out, r, 32);
}

/* [k]P, 4-bit fixed windows (the reference's own native mul is bit-serial: curve/ed25519.rs:55-72) */
static ge ge_scalarmult(const u8 *k, ge p) {
    ge tab[16];
    tab[0] = GE_ID;
    for (int i = 1; i < 16; i++) tab[i] = ge_add(tab[i - 1], p);
    ge r = GE_ID;
    for (int i = 63; i >= 0; i--) {
        r = ge_dbl(ge_dbl(ge_dbl(ge_dbl(r))));
        r = ge_add(r, tab[(k[i >> 1] >> ((i & 1) * 4)) & 15]);
    }
    return r;
}
static ge ge_base(void) {
    static const u8 by[32] = {0x58, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66,
                              0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66};
    ge b;
    ge_decompress(&b, by);
    return b;
}

int zklc_oracle_ed25519_verify(const u8 *pk, const u8 *sig, const u8 *msg, u64 msg_len) {
    if (!sc_lt_l(sig + 32)) return 0;
    ge A;
    if (!ge_decompress(&A, pk)) return 0;
    u8 dg[64], h[32];
    sha512_ctx c;
    sha512_init(&c);
    sha512_update(&c, sig, 32);
    sha512_update(&c, pk, 32);
    sha512_update(&c, msg, msg_len);
    sha512_final(&c, dg);
    sc_reduce64(h, dg);
    A.X = fe_neg(A.X);
    A.T = fe_neg(A.T);
    ge R = ge_add(ge_scalarmult(sig + 32, ge_base()), ge_scalarmult(h, A));
    u8 rc[32];
    ge_compress(rc, R);
    return memcmp(rc, sig, 32) == 0;
}

/* batch; msg_stride == 0 -> shared message.  Returns the thread count used. */
int zklc_oracle_ed25519_verify_batch(const u8 *pks, const u8 *sigs, const u8 *msgs, uint32_t msg_len, uint32_t msg_stride,
                                     uint32_t n, u8 *ok, int nthreads) {
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n; i++)
        ok[i] = (u8)zklc_oracle_ed25519_verify(pks + 32 * i, sigs + 64 * i, msgs + (u64)msg_stride * i, msg_len);
    return used;
}
