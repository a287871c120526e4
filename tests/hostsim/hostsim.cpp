// TEST INFRASTRUCTURE: g++ build of the *.cuh arithmetic headers so the CPU
// test-suite (pytest -m "not gpu") can check the exact kernel arithmetic
// against the oracle without a GPU.  Never linked into libzklc_mi355.so.
#include "../../zk-light-client-implementation_amd/csrc/ed25519_verify.cuh"
#include "../../zk-light-client-implementation_amd/csrc/poseidon_gl.cuh"
#include <string.h>

static ge_niels g_btab[ZKLC_ED_BTABLE];
static int g_btab_ready = 0;

extern "C" {

void hostsim_fe_op(int op, const u32 *a, const u32 *b, u32 *out) {
    fe x = fe_from_words(a), y = fe_from_words(b), r;  // inputs < 2^255
    switch (op) {
        case 0: r = fe_add(x, y); break;
        case 1: r = fe_sub(x, y); break;
        case 2: r = fe_mul(x, y); break;
        case 3: r = fe_sqr(x); break;
        case 4: r = fe_invert(x); break;
        case 5: r = fe_pow22523(x); break;
        case 6: r = fe_freeze(x); break;
        case 7: r = fe_sqr2(x); break;
        case 8: {  // three-term lazy inputs built from REDUCED values
            fe xr = fe_mul(x, fe_one()), yr = fe_mul(y, fe_one());
            r = fe_mul(fe_add(fe_add(xr, yr), xr), fe_sub(fe_sub(yr, xr), xr));
            break;
        }
        default: r = fe_zero();
    }
    fe_freeze_words(out, r);  // canonical
}

void hostsim_sc_reduce512(const u32 *x, u32 *out) { sc_reduce512(out, x); }
u32 hostsim_sc_is_canonical(const u32 *x) { return sc_is_canonical(x); }

void hostsim_sha512(const uint8_t *msg, u32 len, uint8_t *out) {
    u64 h[8];
    sha512_hash_msg(msg, len, h);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(h[i] >> (56 - 8 * j));
}

u32 hostsim_decompress_compress(const uint8_t *in, uint8_t *out) {
    u32 w[8];
    memcpy(w, in, 32);
    ge_p3 p;
    u32 ok = ge_decompress(p, w);
    u32 o[8];
    ge_compress(o, p);
    memcpy(out, o, 32);
    return ok;
}

void hostsim_base_table(u32 *out /*128*30 words*/) {
    if (!g_btab_ready) {
        for (u32 j = 1; j <= ZKLC_ED_BTABLE; j++) g_btab[j - 1] = ed25519_base_table_entry(j);
        g_btab_ready = 1;
    }
    memcpy(out, g_btab, sizeof(g_btab));
}

u32 hostsim_ed25519_verify(const uint8_t *pk, const uint8_t *sig, const uint8_t *msg, u32 msg_len) {
    if (!g_btab_ready) {
        u32 tmp[ZKLC_ED_BTABLE * 30];
        hostsim_base_table(tmp);
    }
    u32 pkw[8], sigw[16];
    memcpy(pkw, pk, 32);
    memcpy(sigw, sig, 64);
    i32 tab[ZKLC_ED_ATAB_WORDS];
    return ed25519_verify_one<1>(pkw, sigw, msg, msg_len, g_btab, tab);
}

u64 hostsim_gl_op(int op, u64 a, u64 b) {
    switch (op) {
        case 0: return gl_add(a, b);
        case 1: return gl_sub(a, b);
        case 2: return gl_mul(a, b);
        case 3: return gl_inv(a);
        case 4: return gl_reduce128(a, b);
        case 5: return gl_root_of_unity((u32)a);
        default: return 0;
    }
}
void hostsim_poseidon_gl_permute(u64 *s) { poseidon_gl_permute(s); }
void hostsim_poseidon_gl_hash(const u64 *in, u32 len, u64 *out4) { poseidon_gl_hash_or_noop(in, 1, len, out4); }
void hostsim_poseidon_gl_two_to_one(const u64 *l, const u64 *r, u64 *out4) { poseidon_gl_two_to_one(l, r, out4); }
}
