/* BN254 G1 MSM -- plain C restatement.  TEST INFRASTRUCTURE / CPU BASELINE ONLY.
 *
 * Independent of the GPU code on purpose: gnark-crypto's own representation (4 x u64
 * Montgomery limbs, R = 2^256, CIOS multiplication with unsigned __int128), Jacobian
 * coordinates, unsigned fixed windows.  Curve and field from
 * contracts/hardhat/contracts/Verifier.sol:28-40; MSM definition sum s_i P_i (gnark-crypto is
 * un-vendored: PARITY UNPINNED, see oracle/bn254.py).  Pinned against oracle/bn254.py in tests.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 PM[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 PINV = 0x87d20782e4866389ULL;                 /* -p^-1 mod 2^64 */
static const u64 ONE_M[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}; /* 2^256 mod p */
typedef struct { u64 v[4]; } fpe;

static int fp_geq_p(const u64 *a) {
    for (int i = 3; i >= 0; i--) { if (a[i] > PM[i]) return 1; if (a[i] < PM[i]) return 0; }
    return 1;
}
static void fp_sub_p(u64 *a) { u128 b = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - PM[i] - (u64)b; a[i] = (u64)d; b = (d >> 64) & 1; } }
static fpe fp_add(fpe a, fpe b) {
    fpe r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (u64)c; c >>= 64; }
    if (c || fp_geq_p(r.v)) fp_sub_p(r.v);
    return r;
}
static fpe fp_sub(fpe a, fpe b) {
    fpe r; u128 bw = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.v[i] - b.v[i] - (u64)bw; r.v[i] = (u64)d; bw = (d >> 64) & 1; }
    if (bw) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.v[i] + PM[i]; r.v[i] = (u64)c; c >>= 64; } }
    return r;
}
static fpe fp_mul(fpe a, fpe b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
        u64 m = t[0] * PINV;
        c = (u128)m * PM[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * PM[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
        c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
    }
    fpe r; memcpy(r.v, t, 32);
    if (t[4] || fp_geq_p(r.v)) fp_sub_p(r.v);
    return r;
}
static int fp_is_zero(fpe a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static int fp_eq(fpe a, fpe b) { return memcmp(a.v, b.v, 32) == 0; }
static fpe fp_inv(fpe a) {
    static const u64 E[4] = {0x3c208c16d87cfd45ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    fpe r; memcpy(r.v, ONE_M, 32);
    for (int i = 253; i >= 0; i--) { r = fp_mul(r, r); if ((E[i >> 6] >> (i & 63)) & 1) r = fp_mul(r, a); }
    return r;
}

typedef struct { fpe X, Y, Z; } jac; /* Z = 0: infinity */
static jac jac_inf(void) { jac r; memset(&r, 0, sizeof r); return r; }
static jac jac_dbl(jac p) {
    if (fp_is_zero(p.Z)) return p;
    fpe A = fp_mul(p.X, p.X), B = fp_mul(p.Y, p.Y), C = fp_mul(B, B);
    fpe t = fp_add(p.X, B); t = fp_mul(t, t); t = fp_sub(fp_sub(t, A), C);
    fpe D = fp_add(t, t), E = fp_add(fp_add(A, A), A), F = fp_mul(E, E);
    jac r;
    r.X = fp_sub(F, fp_add(D, D));
    fpe C8 = fp_add(C, C); C8 = fp_add(C8, C8); C8 = fp_add(C8, C8);
    r.Y = fp_sub(fp_mul(E, fp_sub(D, r.X)), C8);
    fpe yz = fp_mul(p.Y, p.Z);
    r.Z = fp_add(yz, yz);
    return r;
}
static jac jac_add(jac p, jac q) {
    if (fp_is_zero(p.Z)) return q;
    if (fp_is_zero(q.Z)) return p;
    fpe Z1Z1 = fp_mul(p.Z, p.Z), Z2Z2 = fp_mul(q.Z, q.Z);
    fpe U1 = fp_mul(p.X, Z2Z2), U2 = fp_mul(q.X, Z1Z1);
    fpe S1 = fp_mul(fp_mul(p.Y, q.Z), Z2Z2), S2 = fp_mul(fp_mul(q.Y, p.Z), Z1Z1);
    if (fp_eq(U1, U2)) { if (fp_eq(S1, S2)) return jac_dbl(p); return jac_inf(); }
    fpe H = fp_sub(U2, U1), R = fp_sub(S2, S1);
    fpe HH = fp_mul(H, H), HHH = fp_mul(H, HH), V = fp_mul(U1, HH);
    jac r;
    r.X = fp_sub(fp_sub(fp_mul(R, R), HHH), fp_add(V, V));
    r.Y = fp_sub(fp_mul(R, fp_sub(V, r.X)), fp_mul(S1, HHH));
    r.Z = fp_mul(fp_mul(p.Z, q.Z), H);
    return r;
}
static jac jac_from_affine(const u64 *pt) { /* gnark layout; (0,0) = infinity */
    jac r; memcpy(r.X.v, pt, 32); memcpy(r.Y.v, pt + 4, 32);
    if (fp_is_zero(r.X) && fp_is_zero(r.Y)) return jac_inf();
    memcpy(r.Z.v, ONE_M, 32);
    return r;
}
static int jac_to_affine(jac p, u64 *out) {
    if (fp_is_zero(p.Z)) { memset(out, 0, 64); return 1; }
    fpe zi = fp_inv(p.Z), zi2 = fp_mul(zi, zi);
    fpe x = fp_mul(p.X, zi2), y = fp_mul(p.Y, fp_mul(zi2, zi));
    memcpy(out, x.v, 32); memcpy(out + 4, y.v, 32);
    return 0;
}

/* points P_i = base + i * step (i < n) in affine gnark layout, normalised in batches (Montgomery trick).
 * base/step are small multiples of the generator (1, 2): base = a*G, step = b*G. */
static jac jac_mul_small(jac p, u64 k) { jac r = jac_inf(); for (int i = 63; i >= 0; i--) { r = jac_dbl(r); if ((k >> i) & 1) r = jac_add(r, p); } return r; }
void zklc_oracle_bn254_gen_points(u64 a, u64 b, u64 n, u64 *out) {
    fpe gx, gy; /* (1, 2) in Montgomery form */
    memcpy(gx.v, ONE_M, 32); gy = fp_add(gx, gx);
    u64 g[8]; memcpy(g, gx.v, 32); memcpy(g + 4, gy.v, 32);
    jac G = jac_from_affine(g), cur = jac_mul_small(G, a), step = jac_mul_small(G, b);
    const u64 CH = 1024;
    jac *buf = malloc(CH * sizeof(jac)); fpe *pre = malloc(CH * sizeof(fpe));
    for (u64 i0 = 0; i0 < n; i0 += CH) {
        u64 m = n - i0 < CH ? n - i0 : CH;
        fpe acc; memcpy(acc.v, ONE_M, 32);
        for (u64 k = 0; k < m; k++) { buf[k] = cur; cur = jac_add(cur, step); pre[k] = acc; acc = fp_mul(acc, buf[k].Z); }
        fpe inv = fp_inv(acc);
        for (u64 k = m; k-- > 0;) {
            fpe zi = fp_mul(inv, pre[k]); inv = fp_mul(inv, buf[k].Z);
            fpe zi2 = fp_mul(zi, zi);
            fpe x = fp_mul(buf[k].X, zi2), y = fp_mul(buf[k].Y, fp_mul(zi2, zi));
            memcpy(out + (i0 + k) * 8, x.v, 32); memcpy(out + (i0 + k) * 8 + 4, y.v, 32);
        }
    }
    free(buf); free(pre);
}

/* naive: sum of double-and-add products (tiny n only) */
int zklc_oracle_bn254_msm_naive(const u64 *points, const u64 *scalars, u64 n, u64 *out) {
    jac acc = jac_inf();
    for (u64 i = 0; i < n; i++) {
        jac p = jac_from_affine(points + 8 * i), r = jac_inf();
        for (int b = 255; b >= 0; b--) { r = jac_dbl(r); if ((scalars[4 * i + (b >> 6)] >> (b & 63)) & 1) r = jac_add(r, p); }
        acc = jac_add(acc, r);
    }
    return jac_to_affine(acc, out);
}

/* bucket method, unsigned c-bit windows, windows processed in parallel (OpenMP) */
int zklc_oracle_bn254_msm(const u64 *points, const u64 *scalars, u64 n, u64 *out, int nthreads, int *threads_used) {
    int c = n >= (1u << 18) ? 15 : n >= (1u << 12) ? 11 : n >= 64 ? 7 : 3;
    int windows = (256 + c - 1) / c;
    jac *wsum = malloc(windows * sizeof(jac));
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int w = 0; w < windows; w++) {
        u64 nb = (1ULL << c) - 1;
        jac *bk = malloc(nb * sizeof(jac));
        for (u64 b = 0; b < nb; b++) bk[b] = jac_inf();
        for (u64 i = 0; i < n; i++) {
            int bit = w * c, wi = bit >> 6, sh = bit & 63;
            u64 x = scalars[4 * i + wi] >> sh;
            if (sh + c > 64 && wi + 1 < 4) x |= scalars[4 * i + wi + 1] << (64 - sh);
            u64 d = x & nb;
            if (bit + c > 256) d &= (1ULL << (256 - bit)) - 1;
            if (d) bk[d - 1] = jac_add(bk[d - 1], jac_from_affine(points + 8 * i));
        }
        jac run = jac_inf(), tot = jac_inf();
        for (u64 b = nb; b-- > 0;) { run = jac_add(run, bk[b]); tot = jac_add(tot, run); }
        wsum[w] = tot;
        free(bk);
    }
    jac acc = jac_inf();
    for (int w = windows - 1; w >= 0; w--) { for (int k = 0; k < c; k++) acc = jac_dbl(acc); acc = jac_add(acc, wsum[w]); }
    free(wsum);
    if (threads_used) *threads_used = used;
    return jac_to_affine(acc, out);
}
