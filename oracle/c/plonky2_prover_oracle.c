/* A COMPLETE plonky2 prover in plain C + OpenMP -- TEST INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/__init__.py).
 *
 * What it is for: `bench.py`'s `cpu_baseline` times one whole proof of a recursion-shaped circuit (2^12..2^14 rows x 135 wires)
 * on the host cores with this file, and the tests require its proof BYTES to equal the Python restatement's
 * (oracle/plonky2_prover.py) and the GPU prover's on the same circuit and witness.
 *
 * What it restates: the prover of the un-vendored fork wormhole-foundation/plonky2-near @ 2244a9d (Cargo.toml:44-47), reached
 * from `data.prove(pw)` at near_bft_finality/src/prove_crypto/ed25519.rs:60,100 and recursion.rs:95 -- plonky2
 * `plonk/prover.rs` (`prove_with_partition_witness`), `plonk/vanishing_poly.rs`, `fri/oracle.rs`, `fri/prover.rs`,
 * `iop/challenger.rs`, `util/serialization.rs` -- in the order of operations of oracle/plonky2_prover.py, with the gate
 * evaluators of gnark-plonky2-verifier/plonk/gates/ (file:line list in oracle/plonky2_gates.py) for the gate set of the
 * reference's recursion circuits -- Noop, Constant, PublicInput, Arithmetic, ArithmeticExtension, MulExtension, BaseSum,
 * Poseidon, PoseidonMds, RandomAccess, Reducing, ReducingExtension, Exponentiation, CosetInterpolation -- and the in-tree gates
 * of the Ed25519 / SHA circuits (crypto/plonky2_u32/src/gates: U32Arithmetic, U32AddMany, U32Subtraction, U32RangeCheck,
 * Comparison, U32Interleave, UninterleaveToU32 / ToB32), i.e. every gate type of include/zklc.h.
 * PARITY against the Rust prover's bytes is UNPINNED like the Python restatement's (same deterministic choices: lowest
 * proof-of-work witness); every proof is accepted by oracle/plonky2_verifier.py, which the reference's golden proofs pin.
 *
 * Gate codes and parameters are those of include/zklc.h (ZKLC_GATE_*), so that one description of a circuit serves both.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "poseidon_gl_rc.h"
#include "poseidon_gl_fast.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;
#define P 0xFFFFFFFF00000001ULL
#define EPS 0xFFFFFFFFULL
#define GEN 7ULL      /* multiplicative generator = coset shift */
#define W7 7ULL       /* X^2 = 7 */

/* ------------------------------------------------------------------------------------------------ field */
static inline u64 red128(u128 x) {
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;                 /* borrow: + p */
    u64 t1 = hl * EPS;
    u64 r = t0 + t1;
    if (r < t1) r += EPS;                   /* carry: 2^64 = eps */
    return r >= P ? r - P : r;
}
static inline u64 fadd(u64 a, u64 b) { u64 s = a + b; if (s < a || s >= P) s -= P; return s; }
static inline u64 fsub(u64 a, u64 b) { return a >= b ? a - b : a + P - b; }
static inline u64 fmul(u64 a, u64 b) { return red128((u128)a * b); }
static u64 fpow(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = fmul(r, a); a = fmul(a, a); e >>= 1; } return r; }
static inline u64 finv(u64 a) { return fpow(a, P - 2); }
static u64 root_of_unity(int logn) { return fpow(1753635133440165772ULL, 1ULL << (32 - logn)); }

typedef struct { u64 a, b; } e2;
static inline e2 e2_make(u64 a, u64 b) { e2 r = {a, b}; return r; }
static inline e2 e2_add(e2 x, e2 y) { return e2_make(fadd(x.a, y.a), fadd(x.b, y.b)); }
static inline e2 e2_sub(e2 x, e2 y) { return e2_make(fsub(x.a, y.a), fsub(x.b, y.b)); }
static inline e2 e2_mul(e2 x, e2 y) {
    return e2_make(fadd(fmul(x.a, y.a), fmul(W7, fmul(x.b, y.b))), fadd(fmul(x.a, y.b), fmul(x.b, y.a)));
}
static inline e2 e2_scale(e2 x, u64 s) { return e2_make(fmul(x.a, s), fmul(x.b, s)); }
static e2 e2_pow(e2 a, u64 e) { e2 r = {1, 0}; while (e) { if (e & 1) r = e2_mul(r, a); a = e2_mul(a, a); e >>= 1; } return r; }

static inline u64 bitrev(u64 i, int bits) {
    u64 r = 0;
    for (int b = 0; b < bits; b++) r |= ((i >> b) & 1) << (bits - 1 - b);
    return r;
}

/* ------------------------------------------------------------------------------------------------ NTT */
static void ntt_one(u64 *a, int logn, const u64 *tw) {
    u64 n = 1ULL << logn;
    for (u64 i = 0; i < n; i++) { u64 r = bitrev(i, logn); if (r > i) { u64 t = a[i]; a[i] = a[r]; a[r] = t; } }
    for (int s = 0; s < logn; s++) {
        u64 m = 1ULL << s, step = n >> (s + 1);
        for (u64 k = 0; k < n; k += 2 * m)
            for (u64 j = 0; j < m; j++) {
                u64 t = fmul(tw[j * step], a[k + j + m]), u = a[k + j];
                a[k + j] = fadd(u, t);
                a[k + j + m] = fsub(u, t);
            }
    }
}
/* batch transforms, poly-major, natural order in and out; inverse includes 1/n */
static void ntt_batch(u64 *data, int logn, u32 batch, int inverse) {
    u64 n = 1ULL << logn, w = root_of_unity(logn);
    if (inverse) w = finv(w);
    u64 *tw = malloc((n / 2 + 1) * 8);
    tw[0] = 1;
    for (u64 i = 1; i < n / 2; i++) tw[i] = fmul(tw[i - 1], w);
    u64 ninv = finv(n % P);
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < (int64_t)batch; b++) {
        u64 *a = data + (u64)b * n;
        ntt_one(a, logn, tw);
        if (inverse) for (u64 i = 0; i < n; i++) a[i] = fmul(a[i], ninv);
    }
    free(tw);
}
/* out[b][k] = poly_b(shift w_N^k); coeffs [batch][n_in] -> out [batch][N] */
static void coset_lde(const u64 *coeffs, u64 n_in, int logN, u32 batch, u64 shift, u64 *out) {
    u64 N = 1ULL << logN;
    u64 *pw = malloc(n_in * 8);
    pw[0] = 1;
    for (u64 i = 1; i < n_in; i++) pw[i] = fmul(pw[i - 1], shift);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)batch; b++) {
        for (u64 i = 0; i < n_in; i++) out[(u64)b * N + i] = fmul(coeffs[(u64)b * n_in + i], pw[i]);
        memset(out + (u64)b * N + n_in, 0, (N - n_in) * 8);
    }
    free(pw);
    ntt_batch(out, logN, batch, 0);
}

/* ------------------------------------------------------------------------------------------------ Poseidon (textbook rounds) */
static const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static inline u64 sbox7(u64 x) { u64 x2 = fmul(x, x), x3 = fmul(x2, x), x4 = fmul(x2, x2); return fmul(x3, x4); }
/* MDS layer: the entries are 6-bit, so the 32-bit halves of the state are multiplied and summed separately in 64-bit lanes
 * (no 128-bit products, the loops vectorise) and recombined with ONE reduction per output -- plonky2's own CPU trick */
static void mds(u64 *s) {
    u64 lo[24], hi[24], o[12];
    for (int i = 0; i < 12; i++) { lo[i] = lo[i + 12] = (u32)s[i]; hi[i] = hi[i + 12] = s[i] >> 32; }
    for (int r = 0; r < 12; r++) {
        u64 a = 0, b = 0;
        for (int i = 0; i < 12; i++) { a += lo[i + r] * CIRC[i]; b += hi[i + r] * CIRC[i]; }
        if (r == 0) { a += lo[0] * 8; b += hi[0] * 8; }
        o[r] = red128((u128)a + ((u128)b << 32));
    }
    memcpy(s, o, sizeof o);
}
/* sum_i x[i] * k[i] mod p for n <= 16 terms: the constants split into 32-bit halves so that both partial sums fit 128 bits */
static inline u64 dot64(const u64 *x, const unsigned long long *k, int n) {
    u128 lo = 0, hi = 0;
    for (int i = 0; i < n; i++) { lo += (u128)x[i] * (u32)k[i]; hi += (u128)x[i] * (u32)(k[i] >> 32); }
    return fadd(red128(lo), fmul(red128(hi), 1ULL << 32));
}
/* the permutation with the optimised partial rounds (poseidon/goldilocks.go:102-115,231-331): same function as the textbook
 * form of goldilocks_oracle.c -- every Merkle cap and challenge of a proof depends on that */
static void permute(u64 *s) {
    int rnd = 0;
    for (int r = 0; r < 4; r++, rnd++) { for (int i = 0; i < 12; i++) s[i] = sbox7(fadd(s[i], ORACLE_PGL_RC[12 * rnd + i])); mds(s); }
    {
        u64 t[12];
        for (int i = 0; i < 12; i++) s[i] = fadd(s[i], ORACLE_PGL_FP_FIRST[i]);
        t[0] = s[0];
        for (int d = 1; d < 12; d++) {
            u128 lo = 0, hi = 0;
            for (int r = 1; r < 12; r++) {
                unsigned long long k = ORACLE_PGL_FP_INIT[(r - 1) * 11 + d - 1];
                lo += (u128)s[r] * (u32)k;
                hi += (u128)s[r] * (u32)(k >> 32);
            }
            t[d] = fadd(red128(lo), fmul(red128(hi), 1ULL << 32));
        }
        memcpy(s, t, sizeof t);
    }
    for (int r = 0; r < 22; r++) {
        u64 s0 = sbox7(s[0]);
        if (r < 21) s0 = fadd(s0, ORACLE_PGL_FP_RC[r]);
        u64 d = fadd(fmul(s0, 25), dot64(s + 1, ORACLE_PGL_FP_WHATS + 11 * r, 11));
        for (int i = 1; i < 12; i++) s[i] = fadd(s[i], fmul(s0, ORACLE_PGL_FP_VS[11 * r + i - 1]));
        s[0] = d;
    }
    rnd += 22;
    for (int r = 0; r < 4; r++, rnd++) { for (int i = 0; i < 12; i++) s[i] = sbox7(fadd(s[i], ORACLE_PGL_RC[12 * rnd + i])); mds(s); }
}
static void hash_no_pad(const u64 *in, u64 len, u64 *out4) {
    u64 s[12] = {0};
    for (u64 off = 0; off < len; off += 8) {
        for (u64 j = 0; j < 8 && off + j < len; j++) s[j] = in[off + j];
        permute(s);
    }
    memcpy(out4, s, 32);
}
static void hash_or_noop(const u64 *in, u64 len, u64 *out4) {
    if (len <= 4) { for (u64 i = 0; i < 4; i++) out4[i] = i < len ? in[i] : 0; return; }
    hash_no_pad(in, len, out4);
}
static void two_to_one(const u64 *l, const u64 *r, u64 *out4) {
    u64 s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    permute(s);
    memcpy(out4, s, 32);
}

/* ------------------------------------------------------------------------------------------------ Merkle tree */
typedef struct {
    int log_leaves, cap_height;
    u64 *digests;        /* levels concatenated: 2^log_leaves, 2^(log_leaves-1), .. , 2^cap_height hashes of 4 u64 */
} tree_t;
static u64 *tree_level(const tree_t *t, int l) {      /* l = 0: leaf digests */
    u64 off = 0;
    for (int k = 0; k < l; k++) off += 4ULL << (t->log_leaves - k);
    return t->digests + off;
}
static void tree_build(tree_t *t, const u64 *leaves, u64 width, int log_leaves, int cap_height) {
    if (cap_height > log_leaves) cap_height = log_leaves;
    t->log_leaves = log_leaves;
    t->cap_height = cap_height;
    u64 n = 1ULL << log_leaves;
    t->digests = malloc(8 * n * 8);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) hash_or_noop(leaves + (u64)i * width, width, t->digests + 4 * i);
    for (int l = 0; l < log_leaves - cap_height; l++) {
        u64 *lev = tree_level(t, l), *nxt = tree_level(t, l + 1);
        u64 parents = n >> (l + 1);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)parents; i++) two_to_one(lev + 8 * i, lev + 8 * i + 4, nxt + 4 * i);
    }
}
static const u64 *tree_cap(const tree_t *t) { return tree_level(t, t->log_leaves - t->cap_height); }

/* ------------------------------------------------------------------------------------------------ polynomial batch */
typedef struct {
    u32 width;
    u64 n, N;
    u64 *coeffs;   /* [width][n] */
    u64 *lde;      /* [width][N], natural order on g <w_N> */
    u64 *leaves;   /* [N][width], leaf i = the values at natural index bitrev(i) */
    tree_t tree;
} batch_t;
static void batch_from_coeffs(batch_t *b, u64 *coeffs /* owned */, u32 width, int degree_bits, int rate_bits, int cap_height) {
    b->width = width;
    b->n = 1ULL << degree_bits;
    b->N = b->n << rate_bits;
    b->coeffs = coeffs;
    b->lde = malloc((u64)width * b->N * 8);
    coset_lde(coeffs, b->n, degree_bits + rate_bits, width, GEN, b->lde);
    b->leaves = malloc((u64)width * b->N * 8);
    int bits = degree_bits + rate_bits;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)b->N; i++) {
        u64 src = bitrev((u64)i, bits);
        for (u32 k = 0; k < width; k++) b->leaves[(u64)i * width + k] = b->lde[(u64)k * b->N + src];
    }
    tree_build(&b->tree, b->leaves, width, bits, cap_height);
}
static void batch_from_values(batch_t *b, const u64 *values, u32 width, int degree_bits, int rate_bits, int cap_height) {
    u64 n = 1ULL << degree_bits;
    u64 *c = malloc((u64)width * n * 8);
    memcpy(c, values, (u64)width * n * 8);
    ntt_batch(c, degree_bits, width, 1);
    batch_from_coeffs(b, c, width, degree_bits, rate_bits, cap_height);
}
static void batch_free(batch_t *b) { free(b->coeffs); free(b->lde); free(b->leaves); free(b->tree.digests); memset(b, 0, sizeof *b); }

/* ------------------------------------------------------------------------------------------------ challenger */
typedef struct { u64 state[12], inp[8], out[8]; int n_in, n_out; } chal_t;
static void ch_duplex(chal_t *c) {
    for (int i = 0; i < c->n_in; i++) c->state[i] = c->inp[i];
    c->n_in = 0;
    permute(c->state);
    memcpy(c->out, c->state, 64);
    c->n_out = 8;
}
static void ch_observe(chal_t *c, u64 e) {
    c->n_out = 0;
    c->inp[c->n_in++] = e % P;
    if (c->n_in == 8) ch_duplex(c);
}
static void ch_observe_many(chal_t *c, const u64 *e, u64 n) { for (u64 i = 0; i < n; i++) ch_observe(c, e[i]); }
static void ch_observe_ext(chal_t *c, e2 x) { ch_observe(c, x.a); ch_observe(c, x.b); }
static u64 ch_challenge(chal_t *c) {
    if (c->n_in || !c->n_out) ch_duplex(c);
    return c->out[--c->n_out];
}
static e2 ch_ext(chal_t *c) { u64 a = ch_challenge(c), b = ch_challenge(c); return e2_make(a, b); }

/* ------------------------------------------------------------------------------------------------ circuit description */
typedef struct { u32 type, p[4], selector_index, group_start, group_end, extra_off; } ogate;
typedef struct {
    u32 degree_bits, num_wires, num_routed_wires, num_constants, num_selectors, num_challenges, rate_bits, cap_height,
        proof_of_work_bits, num_query_rounds, quotient_degree_factor, num_partial_products, num_gate_constraints,
        num_public_inputs, hasher, num_gates, num_arities, arity_bits[8];
} oparams;
enum { G_NOOP = 0, G_CONSTANT, G_PUBLIC_INPUT, G_ARITHMETIC, G_ARITHMETIC_EXT, G_MUL_EXT, G_BASE_SUM, G_POSEIDON, G_POSEIDON_MDS,
       G_RANDOM_ACCESS, G_REDUCING, G_REDUCING_EXT, G_EXPONENTIATION, G_COSET_INTERPOLATION,
       G_U32_ARITHMETIC, G_U32_ADD_MANY, G_U32_SUBTRACTION, G_U32_RANGE_CHECK, G_COMPARISON,
       G_U32_INTERLEAVE, G_UNINTERLEAVE_TO_U32, G_UNINTERLEAVE_TO_B32, G_NUM_TYPES };
#define UNUSED_SELECTOR 0xFFFFFFFFULL

/* ------------------------------------------------------------------------------------------------ gate evaluators (base field)
 * c: the gate's local constants (after the selectors), w: the wires of the row, out: the constraints; returns their number.
 * "alg" = the degree-2 extension algebra over the base field (pairs, X^2 = 7). */
static inline e2 alg(const u64 *w, u32 s) { return e2_make(w[s], w[s + 1]); }
static inline u64 *put2(u64 *out, e2 v) { out[0] = v.a; out[1] = v.b; return out + 2; }

static u32 eval_poseidon(const u64 *w, u64 *out) {
    u64 *o = out;
    u64 swap = w[24];
    *o++ = fmul(swap, fsub(swap, 1));
    u64 st[12];
    for (int i = 0; i < 4; i++) *o++ = fsub(fmul(swap, fsub(w[i + 4], w[i])), w[25 + i]);
    for (int i = 0; i < 4; i++) { st[i] = fadd(w[i], w[25 + i]); st[i + 4] = fsub(w[i + 4], w[25 + i]); }
    for (int i = 8; i < 12; i++) st[i] = w[i];
    int rnd = 0;
    for (int r = 0; r < 4; r++, rnd++) {
        for (int i = 0; i < 12; i++) st[i] = fadd(st[i], ORACLE_PGL_RC[12 * rnd + i]);
        if (r)
            for (int i = 0; i < 12; i++) { u64 sin = w[29 + 12 * (r - 1) + i]; *o++ = fsub(st[i], sin); st[i] = sin; }
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        mds(st);
    }
    /* the 22 partial rounds in the textbook form: constants on all lanes, S-box on lane 0, full MDS (the same S-box inputs as
     * the reference's fast form, poseidon_gate.go:127-160) */
    for (int r = 0; r < 22; r++, rnd++) {
        for (int i = 0; i < 12; i++) st[i] = fadd(st[i], ORACLE_PGL_RC[12 * rnd + i]);
        u64 sin = w[65 + r];
        *o++ = fsub(st[0], sin);
        st[0] = sbox7(sin);
        mds(st);
    }
    for (int r = 0; r < 4; r++, rnd++) {
        for (int i = 0; i < 12; i++) st[i] = fadd(st[i], ORACLE_PGL_RC[12 * rnd + i]);
        for (int i = 0; i < 12; i++) { u64 sin = w[87 + 12 * r + i]; *o++ = fsub(st[i], sin); st[i] = sin; }
        for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
        mds(st);
    }
    for (int i = 0; i < 12; i++) *o++ = fsub(st[i], w[12 + i]);
    return (u32)(o - out);
}

static inline u64 range_product(u64 x, u32 base) {
    u64 acc = 1;
    for (u32 k = 0; k < base; k++) acc = fmul(acc, fsub(x, k));
    return acc;
}

static void coset_partial(const u64 *w, const u64 *weights, const u64 *dom, u32 s, u32 e, e2 point, e2 *ev, e2 *prod) {
    for (u32 i = s; i < e; i++) {
        e2 term = e2_sub(point, e2_make(dom[i], 0));
        e2 wv = e2_scale(alg(w, 1 + 2 * i), weights[i]);
        *ev = e2_add(e2_mul(*ev, term), e2_mul(wv, *prod));
        *prod = e2_mul(*prod, term);
    }
}

static int eval_gate(const ogate *g, const u64 *c, const u64 *w, const u64 *pih, const u64 *extra, u64 *out) {
    u64 *o = out;
    switch (g->type) {
    case G_NOOP: return 0;
    case G_CONSTANT:
        for (u32 i = 0; i < g->p[0]; i++) *o++ = fsub(c[i], w[i]);
        break;
    case G_PUBLIC_INPUT:
        for (u32 i = 0; i < 4; i++) *o++ = fsub(w[i], pih[i]);
        break;
    case G_ARITHMETIC:
        for (u32 i = 0; i < g->p[0]; i++)
            *o++ = fsub(w[4 * i + 3], fadd(fmul(fmul(w[4 * i], w[4 * i + 1]), c[0]), fmul(w[4 * i + 2], c[1])));
        break;
    case G_ARITHMETIC_EXT:
        for (u32 i = 0; i < g->p[0]; i++) {
            e2 m0 = alg(w, 8 * i), m1 = alg(w, 8 * i + 2), a = alg(w, 8 * i + 4), ov = alg(w, 8 * i + 6);
            e2 comp = e2_add(e2_scale(a, c[1]), e2_scale(e2_mul(m0, m1), c[0]));
            o = put2(o, e2_sub(ov, comp));
        }
        break;
    case G_MUL_EXT:
        for (u32 i = 0; i < g->p[0]; i++) {
            e2 m0 = alg(w, 6 * i), m1 = alg(w, 6 * i + 2), ov = alg(w, 6 * i + 4);
            o = put2(o, e2_sub(ov, e2_scale(e2_mul(m0, m1), c[0])));
        }
        break;
    case G_BASE_SUM: {
        u32 n = g->p[0], base = g->p[1];
        u64 acc = 0;
        for (u32 i = n; i-- > 0;) acc = fadd(fmul(acc, base), w[1 + i]);
        *o++ = fsub(acc, w[0]);
        for (u32 i = 0; i < n; i++) {
            u64 pr = 1;
            for (u32 k = 0; k < base; k++) pr = fmul(pr, fsub(w[1 + i], k));
            *o++ = pr;
        }
        break;
    }
    case G_POSEIDON: return (int)eval_poseidon(w, out);
    case G_POSEIDON_MDS:
        for (u32 r = 0; r < 12; r++) {
            e2 acc = {0, 0};
            for (u32 i = 0; i < 12; i++) acc = e2_add(acc, e2_scale(alg(w, 2 * ((i + r) % 12)), CIRC[i]));
            if (r == 0) acc = e2_add(acc, e2_scale(alg(w, 0), 8));
            o = put2(o, e2_sub(alg(w, 2 * (12 + r)), acc));
        }
        break;
    case G_RANDOM_ACCESS: {
        u32 bits = g->p[0], copies = g->p[1], nextra = g->p[2], vs = 1u << bits;
        u32 routed = (2 + vs) * copies + nextra;
        u64 items[64];
        if (vs > 64) return -1;
        for (u32 cp = 0; cp < copies; cp++) {
            u32 base = (2 + vs) * cp;
            u64 idx = w[base], claimed = w[base + 1];
            const u64 *b = w + routed + cp * bits;
            for (u32 i = 0; i < bits; i++) *o++ = fsub(fmul(b[i], b[i]), b[i]);
            u64 acc = 0;
            for (u32 i = bits; i-- > 0;) acc = fadd(fadd(acc, acc), b[i]);
            *o++ = fsub(acc, idx);
            for (u32 i = 0; i < vs; i++) items[i] = w[base + 2 + i];
            u32 len = vs;
            for (u32 k = 0; k < bits; k++) {
                for (u32 i = 0; i < len; i += 2) items[i / 2] = fadd(items[i], fmul(b[k], fsub(items[i + 1], items[i])));
                len >>= 1;
            }
            *o++ = fsub(items[0], claimed);
        }
        for (u32 i = 0; i < nextra; i++) *o++ = fsub(c[i], w[(2 + vs) * copies + i]);
        break;
    }
    case G_REDUCING:
    case G_REDUCING_EXT: {
        u32 n = g->p[0];
        int ext = g->type == G_REDUCING_EXT;
        e2 alpha = alg(w, 2), acc = alg(w, 4);
        u32 start_accs = 6 + (ext ? 2 * n : n);
        for (u32 i = 0; i < n; i++) {
            e2 nxt = i == n - 1 ? alg(w, 0) : alg(w, start_accs + 2 * i);
            e2 coeff = ext ? alg(w, 6 + 2 * i) : e2_make(w[6 + i], 0);
            o = put2(o, e2_sub(e2_add(e2_mul(acc, alpha), coeff), nxt));
            acc = nxt;
        }
        break;
    }
    case G_EXPONENTIATION: {
        u32 n = g->p[0];
        u64 base = w[0];
        const u64 *bits = w + 1, *inter = w + 2 + n;
        for (u32 i = 0; i < n; i++) {
            u64 prev = i == 0 ? 1 : fmul(inter[i - 1], inter[i - 1]);
            u64 b = bits[n - 1 - i];
            u64 mul_by = fsub(fmul(b, base), fsub(b, 1));
            *o++ = fsub(fmul(prev, mul_by), inter[i]);
        }
        *o++ = fsub(w[1 + n], inter[n - 1]);
        break;
    }
    case G_COSET_INTERPOLATION: {
        u32 sb = g->p[0], d = g->p[1], np = 1u << sb, n_inter = (np - 2) / (d - 1);
        const u64 *weights = extra + g->extra_off, *dom = weights + np;
        u32 start_pt = 1 + 2 * np, start_val = start_pt + 2, start_inter = start_val + 2;
        u64 shift = w[0];
        e2 point = alg(w, start_pt), shifted = alg(w, start_inter + 4 * n_inter);
        o = put2(o, e2_add(e2_scale(shifted, fsub(0, shift)), point));
        e2 ev = {0, 0}, prod = {1, 0};
        coset_partial(w, weights, dom, 0, d, shifted, &ev, &prod);
        for (u32 i = 0; i < n_inter; i++) {
            e2 iev = alg(w, start_inter + 2 * i), ipr = alg(w, start_inter + 2 * (n_inter + i));
            o = put2(o, e2_sub(iev, ev));
            o = put2(o, e2_sub(ipr, prod));
            u32 s = 1 + (d - 1) * (i + 1), e = s + d - 1 < np ? s + d - 1 : np;
            ev = iev;
            prod = ipr;
            coset_partial(w, weights, dom, s, e, shifted, &ev, &prod);
        }
        o = put2(o, e2_sub(alg(w, start_val), ev));
        break;
    }
    /* ---- the in-tree gates of the Ed25519 / SHA circuits (crypto/plonky2_u32/src/gates; list in oracle/plonky2_gates.py) */
    case G_U32_ARITHMETIC: {
        u32 n = g->p[0];
        for (u32 i = 0; i < n; i++) {
            const u64 *r = w + 6 * i, *l = w + 6 * n + 32 * i;
            u64 m0 = r[0], m1 = r[1], add = r[2], lo = r[3], hi = r[4], inv = r[5];
            u64 computed = fadd(fmul(m0, m1), add);
            u64 hi_not_max = fsub(fmul(inv, fsub(0xFFFFFFFFULL, hi)), 1);
            *o++ = fmul(hi_not_max, lo);
            *o++ = fsub(fadd(fmul(hi, 1ULL << 32), lo), computed);
            for (u32 j = 32; j-- > 0;) *o++ = range_product(l[j], 4);
            u64 cl = 0, chh = 0;
            for (u32 j = 16; j-- > 0;) cl = fadd(fmul(cl, 4), l[j]);
            for (u32 j = 32; j-- > 16;) chh = fadd(fmul(chh, 4), l[j]);
            *o++ = fsub(cl, lo);
            *o++ = fsub(chh, hi);
        }
        break;
    }
    case G_U32_ADD_MANY: {
        u32 na = g->p[0], n = g->p[1], per = na + 3;
        for (u32 i = 0; i < n; i++) {
            u64 comp = w[per * i + na];
            for (u32 j = 0; j < na; j++) comp = fadd(comp, w[per * i + j]);
            u64 res = w[per * i + na + 1], carry = w[per * i + na + 2];
            *o++ = fsub(fadd(fmul(carry, 1ULL << 32), res), comp);
            u64 cr = 0, cc = 0;
            for (u32 j = 18; j-- > 0;) {
                u64 l = w[per * n + 18 * i + j];
                *o++ = range_product(l, 4);
                if (j < 16) cr = fadd(fmul(cr, 4), l); else cc = fadd(fmul(cc, 4), l);
            }
            *o++ = fsub(cr, res);
            *o++ = fsub(cc, carry);
        }
        break;
    }
    case G_U32_SUBTRACTION: {
        u32 n = g->p[0];
        for (u32 i = 0; i < n; i++) {
            const u64 *r = w + 5 * i;
            u64 x = r[0], y = r[1], bin = r[2], res = r[3], bout = r[4];
            u64 initial = fsub(fsub(x, y), bin);
            *o++ = fsub(res, fadd(initial, fmul(1ULL << 32, bout)));
            u64 comb = 0;
            for (u32 j = 16; j-- > 0;) {
                u64 l = w[5 * n + 16 * i + j];
                *o++ = range_product(l, 4);
                comb = fadd(fmul(comb, 4), l);
            }
            *o++ = fsub(comb, res);
            *o++ = fmul(bout, fsub(1, bout));
        }
        break;
    }
    case G_U32_RANGE_CHECK: {
        u32 n = g->p[0];
        for (u32 i = 0; i < n; i++) {
            const u64 *aux = w + n + 16 * i;
            u64 acc = 0;
            for (u32 j = 16; j-- > 0;) acc = fadd(fmul(acc, 4), aux[j]);
            *o++ = fsub(acc, w[i]);
            for (u32 j = 0; j < 16; j++) *o++ = range_product(aux[j], 4);
        }
        break;
    }
    case G_COMPARISON: {
        u32 nb = g->p[0], ncn = g->p[1], cb = (nb + ncn - 1) / ncn, size = 1u << cb;
        const u64 *first = w + 4, *second = w + 4 + ncn;
        u64 a = 0, b = 0;
        for (u32 i = ncn; i-- > 0;) { a = fadd(fmul(a, size), first[i]); b = fadd(fmul(b, size), second[i]); }
        *o++ = fsub(a, w[0]);
        *o++ = fsub(b, w[1]);
        u64 msd = 0;
        for (u32 i = 0; i < ncn; i++) {
            *o++ = range_product(first[i], size);
            *o++ = range_product(second[i], size);
            u64 diff = fsub(second[i], first[i]);
            u64 dummy = w[4 + 2 * ncn + i], eq = w[4 + 3 * ncn + i], inter = w[4 + 4 * ncn + i];
            *o++ = fsub(fmul(diff, dummy), fsub(1, eq));
            *o++ = fmul(eq, diff);
            *o++ = fsub(inter, fmul(eq, msd));
            msd = fadd(inter, fmul(fsub(1, eq), diff));
        }
        *o++ = fsub(w[3], msd);
        const u64 *bits = w + 4 + 5 * ncn;
        for (u32 i = 0; i <= cb; i++) *o++ = fmul(bits[i], fsub(1, bits[i]));
        u64 comb = 0;
        for (u32 i = cb + 1; i-- > 0;) comb = fadd(fadd(comb, comb), bits[i]);
        *o++ = fsub(fadd(size, w[3]), comb);
        *o++ = fsub(w[2], bits[cb]);
        break;
    }
    case G_U32_INTERLEAVE: {
        u32 n = g->p[0];
        for (u32 i = 0; i < n; i++) {
            const u64 *bits = w + 2 * n + 32 * i;     /* big-endian */
            u64 x = 0, xi = 0;
            for (u32 j = 0; j < 32; j++) { x = fadd(fadd(x, x), bits[j]); xi = fadd(fmul(xi, 4), bits[j]); }
            *o++ = fsub(x, w[2 * i]);
            *o++ = fsub(xi, w[2 * i + 1]);
            for (u32 j = 0; j < 32; j++) *o++ = range_product(bits[j], 2);
        }
        break;
    }
    case G_UNINTERLEAVE_TO_U32:
    case G_UNINTERLEAVE_TO_B32: {
        u32 n = g->p[0];
        int b32 = g->type == G_UNINTERLEAVE_TO_B32;
        for (u32 i = 0; i < n; i++) {
            const u64 *bits = w + 3 * n + 64 * i;
            u64 x = 0, ev = 0, od = 0;
            for (u32 j = 0; j < 64; j++) x = fadd(fadd(x, x), bits[j]);
            for (u32 j = 0; j < 32; j++) {
                u64 coeff = b32 ? 1ULL << (2 * (31 - j)) : 1ULL << (31 - j);
                ev = fadd(ev, fmul(coeff, bits[2 * j]));
                od = fadd(od, fmul(coeff, bits[2 * j + 1]));
            }
            *o++ = fsub(x, w[3 * i]);
            *o++ = fsub(ev, w[3 * i + 1]);
            *o++ = fsub(od, w[3 * i + 2]);
            for (u32 j = 0; j < 64; j++) *o++ = range_product(bits[j], 2);
        }
        break;
    }
    default: return -1;
    }
    return (int)(o - out);
}

/* ------------------------------------------------------------------------------------------------ proof writer */
typedef struct { uint8_t *p; u64 cap, len; int overflow; } wr_t;
static void wr_u64(wr_t *w, u64 v) { if (w->len + 8 > w->cap) { w->overflow = 1; return; } memcpy(w->p + w->len, &v, 8); w->len += 8; }
static void wr_u8(wr_t *w, uint8_t v) { if (w->len + 1 > w->cap) { w->overflow = 1; return; } w->p[w->len++] = v; }
static void wr_hash(wr_t *w, const u64 *h) { for (int i = 0; i < 4; i++) wr_u64(w, h[i]); }
static void wr_e2(wr_t *w, e2 x) { wr_u64(w, x.a); wr_u64(w, x.b); }
static void wr_merkle_proof(wr_t *w, const tree_t *t, u64 index) {
    int n = t->log_leaves - t->cap_height;
    wr_u8(w, (uint8_t)n);
    for (int l = 0; l < n; l++) {
        wr_hash(w, tree_level(t, l) + 4 * (index ^ 1));
        index >>= 1;
    }
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static e2 eval_poly_ext(const u64 *coeffs, u64 n, e2 x) {
    e2 r = {0, 0};
    for (u64 i = n; i-- > 0;) { r = e2_mul(r, x); r.a = fadd(r.a, coeffs[i]); }
    return r;
}
/* ext coset FFT: coefficients (length len = 2^log_len) -> values on shift <w_len>, natural order */
static void ext_coset_fft(const e2 *coeffs, int log_len, u64 shift, e2 *out) {
    u64 len = 1ULL << log_len;
    u64 *buf = malloc(2 * len * 8);
    u64 s = 1;
    for (u64 i = 0; i < len; i++) { buf[i] = fmul(coeffs[i].a, s); buf[len + i] = fmul(coeffs[i].b, s); s = fmul(s, shift); }
    ntt_batch(buf, log_len, 2, 0);
    for (u64 i = 0; i < len; i++) out[i] = e2_make(buf[i], buf[len + i]);
    free(buf);
}

/* ------------------------------------------------------------------------------------------------ the prover
 * constants: [num_constants][n] (selectors first), sigmas: [num_routed_wires][n], wires: [num_wires][n] -- values on <w_n>.
 * seconds (optional, 8 doubles): preprocessing (constants / sigmas commitment + digest: done once per circuit by any prover),
 * wires commitment, Z + partial products, quotient, openings, FRI, total of the proof proper (all but [0]), threads used.
 * verifier_out (optional): the verifier-only data, 4 * 2^cap_height words of the constants / sigmas cap, then the circuit digest.
 * Returns 0, -1 invalid argument / unsupported gate, -2 the witness does not satisfy the copy constraints, -3 buffer too small. */
int zklc_oracle_plonky2_prove(const oparams *pr, const ogate *gates, const u64 *extra, const u64 *k_is, const u64 *constants,
                              const u64 *sigmas, const u64 *wires, const u64 *public_inputs, uint8_t *proof_out, u64 proof_cap,
                              u64 *proof_len, double *seconds, int nthreads, u64 *verifier_out) {
    if (!pr || !gates || !k_is || !constants || !sigmas || !wires || !proof_out || !proof_len || pr->hasher != 0) return -1;
#ifdef _OPENMP
    omp_set_num_threads(nthreads > 0 ? nthreads : omp_get_num_procs());
#endif
    const int db = (int)pr->degree_bits, rb = (int)pr->rate_bits, lb = db + rb, cap_h = (int)pr->cap_height;
    const u64 n = 1ULL << db, N = 1ULL << lb;
    const u32 nw = pr->num_wires, routed = pr->num_routed_wires, nc = pr->num_constants, nsel = pr->num_selectors;
    const u32 nch = pr->num_challenges, npp = pr->num_partial_products, qdf = pr->quotient_degree_factor;
    const u32 ngc = pr->num_gate_constraints;
    if (qdf != (1u << rb) || nch > 4 || ngc > 1024) return -1;
    for (u32 g = 0; g < pr->num_gates; g++)
        if (gates[g].type >= G_NUM_TYPES) return -1;
    double t[8] = {0}, t0 = now_s();

    /* ---- preprocessing: constants + sigmas commitment, circuit digest = hash(cap || hash_pad([]) || degree_bits) */
    batch_t cs, wb, zb, qb;
    {
        u64 *vals = malloc((u64)(nc + routed) * n * 8);
        memcpy(vals, constants, (u64)nc * n * 8);
        memcpy(vals + (u64)nc * n, sigmas, (u64)routed * n * 8);
        batch_from_values(&cs, vals, nc + routed, db, rb, cap_h);
        free(vals);
    }
    const u32 cap_n = 1u << cs.tree.cap_height;
    u64 digest[4];
    {
        u64 *parts = malloc((4 * cap_n + 5) * 8), pad[8] = {1, 0, 0, 0, 0, 0, 0, 1}, hp[4];
        memcpy(parts, tree_cap(&cs.tree), 4 * cap_n * 8);
        hash_no_pad(pad, 8, hp);
        memcpy(parts + 4 * cap_n, hp, 32);
        parts[4 * cap_n + 4] = (u64)db;
        hash_no_pad(parts, 4 * cap_n + 5, digest);
        free(parts);
    }
    if (verifier_out) {
        memcpy(verifier_out, tree_cap(&cs.tree), 4 * cap_n * 8);
        memcpy(verifier_out + 4 * cap_n, digest, 32);
    }
    t[0] = now_s() - t0;
    double t1 = now_s();

    u64 pih[4];
    hash_no_pad(public_inputs, pr->num_public_inputs, pih);
    chal_t ch;
    memset(&ch, 0, sizeof ch);
    ch_observe_many(&ch, digest, 4);
    ch_observe_many(&ch, pih, 4);

    /* ---- wires */
    batch_from_values(&wb, wires, nw, db, rb, cap_h);
    ch_observe_many(&ch, tree_cap(&wb.tree), 4 * cap_n);
    u64 betas[4], gammas[4], alphas[4];
    for (u32 c = 0; c < nch; c++) betas[c] = ch_challenge(&ch);
    for (u32 c = 0; c < nch; c++) gammas[c] = ch_challenge(&ch);
    t[1] = now_s() - t1;
    t1 = now_s();

    /* ---- Z and partial products: per row and chunk the quotient num / den (one batch inversion per row), then the running product */
    const u64 w_n = root_of_unity(db);
    u64 *sub = malloc(n * 8);
    sub[0] = 1;
    for (u64 i = 1; i < n; i++) sub[i] = fmul(sub[i - 1], w_n);
    const u32 nchunks = npp + 1;
    u64 *zpp = malloc((u64)nch * nchunks * n * 8);       /* [c][0] = Z, [c][1 + k] = partial product k */
    int z_ok = 1;
    for (u32 c = 0; c < nch; c++) {
        u64 *q = malloc((u64)nchunks * n * 8);           /* chunk quotients, [k][i] */
        const u64 b = betas[c], g = gammas[c];
#pragma omp parallel
        {
            u64 *num = malloc(nchunks * 8), *den = malloc(nchunks * 8), *pre = malloc(nchunks * 8);
#pragma omp for schedule(static)
            for (int64_t i = 0; i < (int64_t)n; i++) {
                for (u32 k = 0; k < nchunks; k++) {
                    u64 nu = 1, de = 1;
                    u32 end = (k + 1) * qdf < routed ? (k + 1) * qdf : routed;
                    for (u32 j = k * qdf; j < end; j++) {
                        u64 wv = fadd(wires[(u64)j * n + i], g);
                        nu = fmul(nu, fadd(wv, fmul(b, fmul(k_is[j], sub[i]))));
                        de = fmul(de, fadd(wv, fmul(b, sigmas[(u64)j * n + i])));
                    }
                    num[k] = nu;
                    den[k] = de;
                }
                u64 acc = 1;                               /* batch inversion of the nchunks denominators */
                for (u32 k = 0; k < nchunks; k++) { pre[k] = acc; acc = fmul(acc, den[k]); }
                u64 inv = finv(acc);
                for (u32 k = nchunks; k-- > 0;) { q[(u64)k * n + i] = fmul(num[k], fmul(inv, pre[k])); inv = fmul(inv, den[k]); }
            }
            free(num); free(den); free(pre);
        }
        u64 zx = 1;
        u64 *z = zpp + (u64)c * nchunks * n;
        for (u64 i = 0; i < n; i++) {
            z[i] = zx;
            u64 acc = zx;
            for (u32 k = 0; k < nchunks; k++) {
                acc = fmul(acc, q[(u64)k * n + i]);
                if (k < npp) z[(u64)(1 + k) * n + i] = acc;
            }
            zx = acc;
        }
        if (zx != 1) z_ok = 0;
        free(q);
    }
    if (!z_ok) { free(sub); free(zpp); batch_free(&cs); batch_free(&wb); return -2; }
    {
        /* batch order: Z_0..Z_{nch-1}, then the partial products of challenge 0, 1, .. */
        u64 *vals = malloc((u64)nch * nchunks * n * 8);
        for (u32 c = 0; c < nch; c++) {
            memcpy(vals + (u64)c * n, zpp + (u64)c * nchunks * n, n * 8);
            memcpy(vals + ((u64)nch + (u64)c * npp) * n, zpp + ((u64)c * nchunks + 1) * n, (u64)npp * n * 8);
        }
        batch_from_values(&zb, vals, nch * nchunks, db, rb, cap_h);
        free(vals);
    }
    free(zpp);
    ch_observe_many(&ch, tree_cap(&zb.tree), 4 * cap_n);
    for (u32 c = 0; c < nch; c++) alphas[c] = ch_challenge(&ch);
    t[2] = now_s() - t1;
    t1 = now_s();

    /* ---- quotient on the LDE coset */
    const u32 n_terms = nch + nch * nchunks + ngc;
    u64 *apow = malloc((u64)nch * n_terms * 8);
    for (u32 c = 0; c < nch; c++) { u64 v = 1; for (u32 k = 0; k < n_terms; k++) { apow[(u64)c * n_terms + k] = v; v = fmul(v, alphas[c]); } }
    u64 *qv = malloc((u64)nch * N * 8);
    const u64 w_N = root_of_unity(lb), n_inv_f = finv(n % P);
    u64 *xs = malloc(N * 8);
    xs[0] = GEN;
    for (u64 i = 1; i < N; i++) xs[i] = fmul(xs[i - 1], w_N);
    u64 zh_c[64], zh_inv_c[64];      /* x^n - 1 takes 2^rate_bits values on the coset */
    for (u32 r = 0; r < (1u << rb); r++) { zh_c[r] = fsub(fpow(xs[r], n), 1); zh_inv_c[r] = finv(zh_c[r]); }
    int gate_err = 0;
#pragma omp parallel
    {
        u64 *cw = malloc((u64)(nc + routed + nw + nch * nchunks + nch) * 8), *cons = malloc((ngc + 8) * 8), *gc = malloc((ngc + 8) * 8);
        u64 *terms = malloc((u64)n_terms * 8);
#pragma omp for schedule(dynamic, 64)
        for (int64_t ii = 0; ii < (int64_t)N; ii++) {
            const u64 i = (u64)ii;
            u64 *cv = cw, *sg = cw + nc, *wv = sg + routed, *zp = wv + nw, *zn = zp + nch * nchunks;
            for (u32 k = 0; k < nc + routed; k++) cv[k] = cs.lde[(u64)k * N + i];
            for (u32 k = 0; k < nw; k++) wv[k] = wb.lde[(u64)k * N + i];
            for (u32 k = 0; k < nch * nchunks; k++) zp[k] = zb.lde[(u64)k * N + i];
            const u64 i_next = (i + (1ULL << rb)) & (N - 1);
            for (u32 c = 0; c < nch; c++) zn[c] = zb.lde[(u64)c * N + i_next];
            const u64 x = xs[i], zh = zh_c[i & ((1u << rb) - 1)];
            const u64 l0 = fmul(zh, finv(fmul(fsub(x, 1), n % P)));
            (void)n_inv_f;
            u32 tix = 0;
            for (u32 c = 0; c < nch; c++) terms[tix++] = fmul(l0, fsub(zp[c], 1));
            for (u32 c = 0; c < nch; c++) {
                const u64 b = betas[c], g = gammas[c];
                for (u32 k = 0; k < nchunks; k++) {
                    u64 nu = 1, de = 1;
                    u32 end = (k + 1) * qdf < routed ? (k + 1) * qdf : routed;
                    for (u32 j = k * qdf; j < end; j++) {
                        u64 wg = fadd(wv[j], g);
                        nu = fmul(nu, fadd(fmul(b, fmul(x, k_is[j])), wg));
                        de = fmul(de, fadd(fmul(b, sg[j]), wg));
                    }
                    u64 a0 = k == 0 ? zp[c] : zp[nch + c * npp + k - 1];
                    u64 a1 = k < npp ? zp[nch + c * npp + k] : zn[c];
                    terms[tix++] = fsub(fmul(a0, nu), fmul(a1, de));
                }
            }
            for (u32 k = 0; k < ngc; k++) cons[k] = 0;
            for (u32 g = 0; g < pr->num_gates; g++) {
                const ogate *gt = &gates[g];
                u64 f = 1, s = cv[gt->selector_index];
                for (u32 r = gt->group_start; r < gt->group_end; r++)
                    if (r != g) f = fmul(f, fsub(r, s));
                if (nsel > 1) f = fmul(f, fsub(UNUSED_SELECTOR, s));
                int cnt = eval_gate(gt, cv + nsel, wv, pih, extra, gc);
                if (cnt < 0 || (u32)cnt > ngc) { gate_err = 1; continue; }
                for (int k = 0; k < cnt; k++) cons[k] = fadd(cons[k], fmul(gc[k], f));
            }
            for (u32 k = 0; k < ngc; k++) terms[tix++] = cons[k];
            const u64 zi = zh_inv_c[i & ((1u << rb) - 1)];
            for (u32 c = 0; c < nch; c++) {
                u128 acc = 0;
                u64 hi_acc = 0;
                for (u32 k = 0; k < n_terms; k++) {
                    u128 pr2 = (u128)terms[k] * apow[(u64)c * n_terms + k];
                    acc += pr2;
                    if (acc < pr2) hi_acc++;          /* 2^128 overflow counter */
                }
                u64 r = red128(acc);
                /* 2^128 = -2^32 (mod p) */
                r = fsub(r, fmul(hi_acc % P, 1ULL << 32));
                qv[(u64)c * N + i] = fmul(r, zi);
            }
        }
        free(cw); free(cons); free(gc); free(terms);
    }
    free(apow);
    if (gate_err) { free(qv); free(xs); free(sub); batch_free(&cs); batch_free(&wb); batch_free(&zb); return -1; }
    {
        /* values on g <w_N> -> coefficients (inverse transform, undo the coset shift), split into qdf chunks of n */
        ntt_batch(qv, lb, nch, 1);
        u64 gi = finv(GEN);
        u64 *chunks = malloc((u64)nch * qdf * n * 8);
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < (int64_t)nch; c++) {
            u64 s = 1;
            for (u64 j = 0; j < N; j++) {
                u64 v = fmul(qv[(u64)c * N + j], s);
                s = fmul(s, gi);
                if (j < (u64)qdf * n) chunks[((u64)c * qdf) * n + j] = v;
            }
        }
        batch_from_coeffs(&qb, chunks, nch * qdf, db, rb, cap_h);
    }
    free(qv);
    ch_observe_many(&ch, tree_cap(&qb.tree), 4 * cap_n);
    const e2 zeta = ch_ext(&ch);
    t[3] = now_s() - t1;
    t1 = now_s();

    /* ---- openings */
    batch_t *bs[4] = {&cs, &wb, &zb, &qb};
    u32 n_polys = 0;
    for (int k = 0; k < 4; k++) n_polys += bs[k]->width;
    e2 *open0 = malloc((u64)n_polys * sizeof(e2)), *open1 = malloc((u64)nch * sizeof(e2));
    const e2 g_zeta = e2_scale(zeta, w_n);
    {
        u32 off = 0;
        for (int k = 0; k < 4; k++) {
            batch_t *b = bs[k];
#pragma omp parallel for schedule(dynamic, 1)
            for (int64_t j = 0; j < (int64_t)b->width; j++) open0[off + j] = eval_poly_ext(b->coeffs + (u64)j * n, n, zeta);
            off += b->width;
        }
        for (u32 c = 0; c < nch; c++) open1[c] = eval_poly_ext(zb.coeffs + (u64)c * n, n, g_zeta);
    }
    for (u32 j = 0; j < n_polys; j++) ch_observe_ext(&ch, open0[j]);
    for (u32 c = 0; c < nch; c++) ch_observe_ext(&ch, open1[c]);
    t[4] = now_s() - t1;
    t1 = now_s();

    /* ---- FRI: batched opening quotient in coefficient form, commit phase, proof of work, queries */
    const e2 fri_alpha = ch_ext(&ch);
    e2 *final = calloc(N, sizeof(e2));                    /* n coefficients, zero-padded to N */
    {
        e2 *comp = malloc(n * sizeof(e2)), *q = malloc(n * sizeof(e2));
        for (int part = 0; part < 2; part++) {
            const e2 point = part == 0 ? zeta : g_zeta;
            const u32 cnt = part == 0 ? n_polys : nch;
#pragma omp parallel for schedule(static)
            for (int64_t j = 0; j < (int64_t)n; j++) {
                e2 acc = {0, 0};
                if (part == 0) {
                    for (int k = 3; k >= 0; k--)
                        for (u32 pi = bs[k]->width; pi-- > 0;) {
                            acc = e2_mul(acc, fri_alpha);
                            acc.a = fadd(acc.a, bs[k]->coeffs[(u64)pi * n + j]);
                        }
                } else {
                    for (u32 pi = nch; pi-- > 0;) { acc = e2_mul(acc, fri_alpha); acc.a = fadd(acc.a, zb.coeffs[(u64)pi * n + j]); }
                }
                comp[j] = acc;
            }
            e2 carry = {0, 0};                            /* (comp - comp(point)) / (X - point) */
            for (u64 j = n; j-- > 0;) {
                e2 cur = e2_add(comp[j], e2_mul(carry, point));
                if (j) q[j - 1] = cur;
                carry = cur;
            }
            q[n - 1] = e2_make(0, 0);
            const e2 scale = e2_pow(fri_alpha, cnt);
            for (u64 j = 0; j < n; j++) final[j] = e2_add(e2_mul(final[j], scale), q[j]);
        }
        free(comp); free(q);
    }
    wr_t W = {proof_out, proof_cap, 0, 0};
    for (u32 k = 0; k < 4 * cap_n; k += 4) wr_hash(&W, tree_cap(&wb.tree) + k);
    for (u32 k = 0; k < 4 * cap_n; k += 4) wr_hash(&W, tree_cap(&zb.tree) + k);
    for (u32 k = 0; k < 4 * cap_n; k += 4) wr_hash(&W, tree_cap(&qb.tree) + k);
    {
        /* openings in the order constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys */
        u32 o_cs = 0, o_w = cs.width, o_z = o_w + wb.width, o_q = o_z + zb.width;
        for (u32 j = 0; j < cs.width; j++) wr_e2(&W, open0[o_cs + j]);
        for (u32 j = 0; j < wb.width; j++) wr_e2(&W, open0[o_w + j]);
        for (u32 j = 0; j < nch; j++) wr_e2(&W, open0[o_z + j]);
        for (u32 j = 0; j < nch; j++) wr_e2(&W, open1[j]);
        for (u32 j = nch; j < zb.width; j++) wr_e2(&W, open0[o_z + j]);
        for (u32 j = 0; j < qb.width; j++) wr_e2(&W, open0[o_q + j]);
    }
    const u32 na = pr->num_arities;
    tree_t ftrees[8];
    e2 *fleaves[8];
    int flog[8];
    {
        int log_len = lb;
        e2 *coeffs = final;
        e2 *values = malloc(N * sizeof(e2));
        u64 shift = GEN;
        ext_coset_fft(coeffs, log_len, shift, values);
        for (u32 a = 0; a < na; a++) {
            const int ab = (int)pr->arity_bits[a];
            const u64 len = 1ULL << log_len, arity = 1ULL << ab;
            e2 *rv = malloc(len * sizeof(e2));
            for (u64 i = 0; i < len; i++) rv[i] = values[bitrev(i, log_len)];
            fleaves[a] = rv;
            flog[a] = log_len - ab;
            tree_build(&ftrees[a], (const u64 *)rv, 2 * arity, log_len - ab, cap_h);
            const u32 fcap = 1u << ftrees[a].cap_height;
            ch_observe_many(&ch, tree_cap(&ftrees[a]), 4 * fcap);
            for (u32 k = 0; k < 4 * fcap; k += 4) wr_hash(&W, tree_cap(&ftrees[a]) + k);
            const e2 beta = ch_ext(&ch);
            e2 *nc2 = malloc((len >> ab) * sizeof(e2));
            for (u64 k = 0; k < (len >> ab); k++) {
                e2 acc = {0, 0};
                for (u64 tt = arity; tt-- > 0;) acc = e2_add(e2_mul(acc, beta), coeffs[k * arity + tt]);
                nc2[k] = acc;
            }
            if (coeffs != final) free(coeffs);
            coeffs = nc2;
            log_len -= ab;
            shift = fpow(shift, arity);
            ext_coset_fft(coeffs, log_len, shift, values);
        }
        const u64 flen = (1ULL << log_len) >> rb;
        /* commit-phase caps are written above; the query rounds come before the final polynomial in the byte layout, so the
         * final polynomial and the proof-of-work witness are kept and written after them */
        e2 *final_poly = malloc(flen * sizeof(e2));
        memcpy(final_poly, coeffs, flen * sizeof(e2));
        if (coeffs != final) free(coeffs);
        free(values);
        for (u64 k = 0; k < flen; k++) ch_observe_ext(&ch, final_poly[k]);
        /* proof of work: the lowest witness whose response has pow_bits leading zeros */
        u64 witness = 0;
        {
            u64 base[12];
            memcpy(base, ch.state, sizeof base);
            for (int i = 0; i < ch.n_in; i++) base[i] = ch.inp[i];
            const int slot = ch.n_in;
            const u64 limit = pr->proof_of_work_bits ? (1ULL << (64 - pr->proof_of_work_bits)) : 0;
            u64 found = UINT64_MAX, start = 0;
            while (found == UINT64_MAX) {
                const u64 chunk = 1ULL << 16;
#pragma omp parallel for schedule(static)
                for (int64_t k = 0; k < (int64_t)chunk; k++) {
                    u64 st[12];
                    memcpy(st, base, sizeof st);
                    st[slot] = start + (u64)k;
                    permute(st);
                    if (limit == 0 || st[7] < limit) {
#pragma omp critical
                        if (start + (u64)k < found) found = start + (u64)k;
                    }
                }
                start += chunk;
            }
            witness = found;
        }
        ch_observe(&ch, witness);
        (void)ch_challenge(&ch);
        for (u32 r = 0; r < pr->num_query_rounds; r++) {
            u64 x_index = ch_challenge(&ch) % N;
            for (int k = 0; k < 4; k++) {
                for (u32 j = 0; j < bs[k]->width; j++) wr_u64(&W, bs[k]->leaves[x_index * bs[k]->width + j]);
                wr_merkle_proof(&W, &bs[k]->tree, x_index);
            }
            u64 idx = x_index;
            for (u32 a = 0; a < na; a++) {
                const int ab = (int)pr->arity_bits[a];
                idx >>= ab;
                for (u64 e = 0; e < (1ULL << ab); e++) wr_e2(&W, fleaves[a][(idx << ab) + e]);
                wr_merkle_proof(&W, &ftrees[a], idx);
            }
        }
        for (u64 k = 0; k < flen; k++) wr_e2(&W, final_poly[k]);
        wr_u64(&W, witness);
        free(final_poly);
        (void)flog;
    }
    wr_u64(&W, pr->num_public_inputs);
    for (u32 k = 0; k < pr->num_public_inputs; k++) wr_u64(&W, public_inputs[k]);
    t[5] = now_s() - t1;
    t[6] = t[1] + t[2] + t[3] + t[4] + t[5];
#ifdef _OPENMP
    t[7] = (double)omp_get_max_threads();
#else
    t[7] = 1;
#endif
    if (seconds) memcpy(seconds, t, sizeof t);
    for (u32 a = 0; a < na; a++) { free(ftrees[a].digests); free(fleaves[a]); }
    free(final); free(open0); free(open1); free(xs); free(sub);
    batch_free(&cs); batch_free(&wb); batch_free(&zb); batch_free(&qb);
    *proof_len = W.len;
    return W.overflow ? -3 : 0;
}
