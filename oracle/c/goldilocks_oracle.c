/* Goldilocks NTT / LDE, Poseidon and Merkle tree -- plain C restatement.  TEST
 * INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/__init__.py).
 *
 * Follows gnark-plonky2-verifier/goldilocks/base.go:33-42 (field parameters),
 * poseidon/goldilocks.go:30-37,41-86,92-100,117-125,138-145,172-216 in the TEXTBOOK round
 * structure (round constants on all lanes, S-box on lane 0, full MDS in every partial round
 * -- deliberately not the "fast" form the GPU kernel uses), and fri/fri.go:97-144 for the
 * Merkle conventions.  The NTT is the published definition of plonky2's transform
 * (un-vendored): PARITY UNPINNED, checked for self-consistency only.
 * Pinned by oracle/poseidon_gl.py (itself pinned by the reference's two KATs) in tests.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "poseidon_gl_rc.h"

typedef uint64_t u64;
typedef unsigned __int128 u128;
#define P 0xFFFFFFFF00000001ULL

static inline u64 gl_add(u64 a, u64 b) { u128 s = (u128)a + b; return (u64)(s >= P ? s - P : s); }
static inline u64 gl_sub(u64 a, u64 b) { return a >= b ? a - b : a + P - b; }
static inline u64 gl_mul(u64 a, u64 b) { return (u64)(((u128)a * b) % P); }
static u64 gl_pow(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; } return r; }
u64 zklc_oracle_gl_root(int logn) { return gl_pow(1753635133440165772ULL, 1ULL << (32 - logn)); }

/* in-place natural->natural radix-2 NTT of one polynomial; tw = w^i, i < n/2 */
static void ntt_one(u64 *a, int logn, const u64 *tw) {
    u64 n = 1ULL << logn;
    for (u64 i = 0; i < n; i++) {
        u64 r = 0;
        for (int b = 0; b < logn; b++) r |= ((i >> b) & 1) << (logn - 1 - b);
        if (r > i) { u64 t = a[i]; a[i] = a[r]; a[r] = t; }
    }
    for (int s = 0; s < logn; s++) {
        u64 m = 1ULL << s, step = n >> (s + 1);
        for (u64 k = 0; k < n; k += 2 * m)
            for (u64 j = 0; j < m; j++) {
                u64 t = gl_mul(tw[j * step], a[k + j + m]), u = a[k + j];
                a[k + j] = gl_add(u, t);
                a[k + j + m] = gl_sub(u, t);
            }
    }
}

/* batched transform, poly-major.  inverse includes 1/n.  Returns threads used. */
int zklc_oracle_gl_ntt(u64 *data, int logn, uint32_t batch, int inverse, int nthreads) {
    u64 n = 1ULL << logn;
    u64 w = zklc_oracle_gl_root(logn);
    if (inverse) w = gl_pow(w, P - 2);
    u64 *tw = malloc((n / 2 + 1) * 8);
    tw[0] = 1;
    for (u64 i = 1; i < n / 2; i++) tw[i] = gl_mul(tw[i - 1], w);
    u64 ninv = gl_pow(n % P, P - 2);
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int64_t b = 0; b < (int64_t)batch; b++) {
        u64 *a = data + (u64)b * n;
        ntt_one(a, logn, tw);
        if (inverse) for (u64 i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
    }
    free(tw);
    return used;
}

/* coset LDE: out[b][k] = poly_b(shift * w_N^k), natural order */
int zklc_oracle_gl_lde(const u64 *coeffs, int logn, int rate_bits, uint32_t batch, u64 shift, u64 *out, int nthreads) {
    u64 n = 1ULL << logn, N = n << rate_bits;
    for (uint32_t b = 0; b < batch; b++) {
        u64 s = 1;
        for (u64 i = 0; i < n; i++) { out[(u64)b * N + i] = gl_mul(coeffs[(u64)b * n + i], s); s = gl_mul(s, shift); }
        memset(out + (u64)b * N + n, 0, (N - n) * 8);
    }
    return zklc_oracle_gl_ntt(out, logn + rate_bits, batch, 0, nthreads);
}

/* ---------------- Poseidon (textbook round structure) ---------------- */
static const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static inline u64 sbox(u64 x) { u64 x2 = gl_mul(x, x), x3 = gl_mul(x2, x); return gl_mul(gl_mul(x3, x3), x); }
static void mds(u64 *s) {
    u64 o[12];
    for (int r = 0; r < 12; r++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)s[(i + r) % 12] * CIRC[i];
        if (r == 0) acc += (u128)s[0] * 8;
        o[r] = (u64)(acc % P);
    }
    memcpy(s, o, sizeof o);
}
void zklc_oracle_poseidon_gl_permute(u64 *s) {
    int rnd = 0;
    for (int r = 0; r < 4; r++, rnd++) { for (int i = 0; i < 12; i++) s[i] = sbox(gl_add(s[i], ORACLE_PGL_RC[12 * rnd + i])); mds(s); }
    for (int r = 0; r < 22; r++, rnd++) { for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], ORACLE_PGL_RC[12 * rnd + i]); s[0] = sbox(s[0]); mds(s); }
    for (int r = 0; r < 4; r++, rnd++) { for (int i = 0; i < 12; i++) s[i] = sbox(gl_add(s[i], ORACLE_PGL_RC[12 * rnd + i])); mds(s); }
}
static void hash_or_noop(const u64 *in, u64 stride, uint32_t len, u64 *out4) {
    if (len <= 4) { for (uint32_t i = 0; i < 4; i++) out4[i] = i < len ? in[i * stride] : 0; return; }
    u64 s[12] = {0};
    for (uint32_t off = 0; off < len; off += 8) {
        for (uint32_t j = 0; j < 8 && off + j < len; j++) s[j] = in[(u64)(off + j) * stride];
        zklc_oracle_poseidon_gl_permute(s);
    }
    memcpy(out4, s, 32);
}
static void two_to_one(const u64 *l, const u64 *r, u64 *out4) {
    u64 s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    zklc_oracle_poseidon_gl_permute(s);
    memcpy(out4, s, 32);
}
/* tree layout identical to zklc_gl_merkle_commit: levels concatenated, leaves first, cap last */
int zklc_oracle_gl_merkle_commit(const u64 *mat, u64 stride, int log_leaves, uint32_t width, int cap_height, u64 *tree, int nthreads) {
    u64 n = 1ULL << log_leaves;
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n; i++) hash_or_noop(mat + i, stride, width, tree + 4 * i);
    u64 *level = tree;
    for (int l = 0; l < log_leaves - cap_height; l++) {
        u64 parents = n >> (l + 1);
        u64 *next = level + (4ULL << (log_leaves - l));
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
        for (int64_t i = 0; i < (int64_t)parents; i++) two_to_one(level + 8 * i, level + 8 * i + 4, next + 4 * i);
        level = next;
    }
    return used;
}
