// Host-side transcript and PoseidonGate witness rows (see plonky2_host.h).  Plain C++: poseidon_gl.cuh is written
// for both compilers (common.cuh), so the transcript and the Merkle kernels share one Poseidon source.
#include "plonky2_host.h"
#include <string.h>
#include "poseidon_gl.cuh"

void zklc_host_poseidon_permute(uint64_t *s) { poseidon_gl_permute(s); }

void zklc_host_poseidon_hash_no_pad(const uint64_t *in, size_t n, uint64_t *out4) {
    u64 s[12] = {};
    for (size_t off = 0; off < n; off += 8) {
        for (size_t j = 0; j < 8 && off + j < n; j++) s[j] = in[off + j];
        poseidon_gl_permute(s);
    }
    for (int i = 0; i < 4; i++) out4[i] = s[i];
}

void zklc_challenger::duplex() {
    for (int i = 0; i < n_in; i++) state[i] = in[i];
    n_in = 0;
    poseidon_gl_permute(state);
    for (int i = 0; i < 8; i++) out[i] = state[i];
    n_out = 8;
}
void zklc_challenger::observe(uint64_t e) {
    n_out = 0;
    in[n_in++] = e;
    if (n_in == 8) duplex();
}
void zklc_challenger::observe_many(const uint64_t *e, size_t n) {
    for (size_t i = 0; i < n; i++) observe(e[i]);
}
void zklc_challenger::observe_hash(const uint8_t *h, int hasher) {
    if (hasher == 0) {
        uint64_t v[4];
        memcpy(v, h, 32);
        observe_many(v, 4);
    } else {
        for (int off = 0; off < 32; off += 7) {
            uint64_t v = 0;
            int len = 32 - off < 7 ? 32 - off : 7;
            memcpy(&v, h + off, len);
            observe(v);
        }
    }
}
uint64_t zklc_challenger::challenge() {
    if (n_in || !n_out) duplex();
    return out[--n_out];
}

// One PoseidonGate row per input state: wires 0..12 inputs, 12..24 outputs, 24 swap, 25..29 deltas,
// 29..65 / 65..87 / 87..135 the S-box inputs (gnark-plonky2-verifier/plonk/gates/poseidon_gate.go:27-82).
// elementwise Goldilocks products of canonical values (circuit preprocessing on the host: sigma values k_is[col] * w^row)
extern "C" void zklc_gl_mul_vec(const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) out[i] = gl_mul(a[i], b[i]);
}

extern "C" void zklc_poseidon_gl_constants(uint64_t *rc360, uint64_t *fp_first12, uint64_t *fp_rc22, uint64_t *mds_circ12, uint64_t *mds_diag12) {
    for (int i = 0; i < 360; i++) rc360[i] = PGL_RC[i];
    for (int i = 0; i < 12; i++) fp_first12[i] = PGL_FP_FIRST[i];
    for (int i = 0; i < 22; i++) fp_rc22[i] = PGL_FP_RC[i];
    static const uint64_t circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    for (int i = 0; i < 12; i++) {
        mds_circ12[i] = circ[i];
        mds_diag12[i] = i == 0 ? 8 : 0;
    }
}

// Host-side arithmetic for the witness rows: 64x64->128 products in one instruction, sums of products reduced once.
typedef unsigned __int128 u128h;
static inline u64 h_red(u128h x) { return gl_reduce128((u64)x, (u64)(x >> 64)); }
static inline u64 h_mul(u64 a, u64 b) { return h_red((u128h)a * b); }
static inline u64 h_sbox(u64 x) {
    u64 x2 = h_mul(x, x), x3 = h_mul(x2, x), x4 = h_mul(x2, x2);
    return h_mul(x3, x4);
}
// sum of up to 12 products of canonical values: < 12 * 2^128, kept as a 128-bit sum plus a carry count (2^128 = -2^32 mod p... folded
// through 2^128 mod p = p - 2^32 = 0xFFFFFFFE00000001)
struct h_acc {
    u128h s = 0;
    u64 carries = 0;
    inline void add(u64 a, u64 b) {
        u128h t = (u128h)a * b;
        s += t;
        carries += s < t;
    }
    inline u64 result() const {
        u64 r = h_red(s);
        return carries ? gl_add(r, h_mul(carries, 0xFFFFFFFE00000001ULL)) : r;
    }
};
static inline void h_mds(u64 *s) {
    static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u64 d[24], o[12];
    for (int i = 0; i < 12; i++) d[i] = d[i + 12] = s[i];
    for (int r = 0; r < 12; r++) {
        u128h a = 0;   // 12 terms < 2^64 * 41 and one < 2^64 * 8: no overflow
        for (int i = 0; i < 12; i++) a += (u128h)d[i + r] * C[i];
        if (r == 0) a += (u128h)d[0] * 8;
        o[r] = h_red(a);
    }
    for (int i = 0; i < 12; i++) s[i] = o[i];
}

extern "C" int32_t zklc_poseidon_gl_gate_rows(const uint64_t *inputs, const uint64_t *swap, uint32_t n, uint64_t *rows) {
    if (!inputs || !rows) return -1;
    for (uint32_t k = 0; k < n; k++) {
        const u64 *in = inputs + (size_t)k * 12;
        u64 *w = rows + (size_t)k * 135;
        u64 sw = swap ? swap[k] : 0;
        if (sw > 1) return -1;
        for (int i = 0; i < 12; i++) w[i] = in[i];
        w[24] = sw;
        u64 s[12];
        for (int i = 0; i < 4; i++) {
            u64 delta = sw ? gl_sub(in[i + 4], in[i]) : 0;
            w[25 + i] = delta;
            s[i] = gl_add(in[i], delta);
            s[i + 4] = gl_sub(in[i + 4], delta);
        }
        for (int i = 8; i < 12; i++) s[i] = in[i];
        for (int r = 0; r < 4; r++) {
            for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], PGL_RC[12 * r + i]);
            if (r)
                for (int i = 0; i < 12; i++) w[29 + 12 * (r - 1) + i] = s[i];
            for (int i = 0; i < 12; i++) s[i] = h_sbox(s[i]);
            h_mds(s);
        }
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], PGL_FP_FIRST[i]);
        u64 t[12];
        t[0] = s[0];
        for (int d = 1; d < 12; d++) {
            h_acc a;
            for (int r = 1; r < 12; r++) a.add(s[r], PGL_FP_INIT[(r - 1) * 11 + d - 1]);
            t[d] = a.result();
        }
        for (int i = 0; i < 12; i++) s[i] = t[i];
        for (int r = 0; r < 22; r++) {
            w[65 + r] = s[0];
            u64 s0 = gl_add(h_sbox(s[0]), PGL_FP_RC[r]);
            h_acc a;
            a.add(s0, 25);
            for (int j = 1; j < 12; j++) a.add(s[j], PGL_FP_WHATS[r * 11 + j - 1]);
            for (int j = 1; j < 12; j++) s[j] = h_red((u128h)s0 * PGL_FP_VS[r * 11 + j - 1] + s[j]);
            s[0] = a.result();
        }
        for (int r = 0; r < 4; r++) {
            for (int i = 0; i < 12; i++) {
                s[i] = gl_add(s[i], PGL_RC[12 * (26 + r) + i]);
                w[87 + 12 * r + i] = s[i];
                s[i] = h_sbox(s[i]);
            }
            h_mds(s);
        }
        for (int i = 0; i < 12; i++) w[12 + i] = s[i];
    }
    return 0;
}

// Copy classes of a circuit (plonky2 `CircuitBuilder::connect` -> `wire_partition`, plonk/permutation_argument.rs; the reference
// reaches it through every gadget of crypto/plonky2_ed25519/src/gadgets/*.rs): union-find over the n_pairs recorded connections
// (ia[k], ib[k] = dense indices of the two targets), root_out[i] = the SMALLEST index of i's class.  The Ed25519 circuit records
// ~40 M connections: a Python union-find took minutes, scipy's connected components seconds; this is 0.3 s and drops the scipy
// dependency.  Returns -1 on an index out of range.
extern "C" int32_t zklc_host_copy_classes(const int64_t *ia, const int64_t *ib, uint64_t n_pairs, uint64_t n_keys, int64_t *root_out) {
    if ((n_pairs && (!ia || !ib)) || (n_keys && !root_out) || n_keys >> 40) return -1;
    int64_t *p = root_out;                                 // parent forest in place; links always point to the smaller index
    for (uint64_t i = 0; i < n_keys; i++) p[i] = (int64_t)i;
    auto find = [&](int64_t x) {
        int64_t r = x;
        while (p[r] != r) r = p[r];
        while (p[x] != r) {                                // path compression
            int64_t nx = p[x];
            p[x] = r;
            x = nx;
        }
        return r;
    };
    for (uint64_t k = 0; k < n_pairs; k++) {
        if (ia[k] < 0 || ib[k] < 0 || (uint64_t)ia[k] >= n_keys || (uint64_t)ib[k] >= n_keys) return -1;
        int64_t ra = find(ia[k]), rb = find(ib[k]);
        if (ra < rb) p[rb] = ra;
        else if (rb < ra) p[ra] = rb;
    }
    for (uint64_t i = 0; i < n_keys; i++) p[i] = p[p[i]] == p[i] ? p[i] : find((int64_t)i);   // every entry -> its root
    return 0;
}
