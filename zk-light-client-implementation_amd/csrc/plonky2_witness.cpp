// Native witness generation for circuits built by zklc_amd.plonky2.CircuitBuilder (host code, plain C++).
//
// Replaces the witness generators that run inside `CircuitData::prove` of the reference:
//   crypto/plonky2_ed25519/src/gadgets/nonnative.rs:447-705  NonNative{Addition,Subtraction,Multiplication,Inverse}Generator
//   crypto/plonky2_ed25519/src/gadgets/curve.rs:327-370      CurvePointDecompressionGenerator
//   crypto/plonky2_ecdsa/src/gadgets/biguint.rs:417-470      BigUintDivRemGenerator
//   crypto/plonky2_u32/src/gates/*.rs `generators()`          U32 arithmetic / add-many / subtraction / range-check /
//                                                             comparison gate generators
//   plonky2 (un-vendored) ArithmeticGate / BaseSumGate / RandomAccessGate / PoseidonGate / EqualityGenerator
// The builder records one instruction per generator (opcode, parameters, input slots, output slots; a slot = one copy
// class of the circuit); this interpreter executes the program for a batch of partial witnesses, one host thread per
// witness, and scatters the slot values into the poly-major wire matrix the prover takes.  Writing a slot twice with
// different values is the reference's "copy constraint violated" failure (e.g. an invalid signature) -> error.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>
#include "plonky2_host.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;
static const u64 GLP = 0xFFFFFFFF00000001ULL;

static inline u64 g_mul(u64 a, u64 b) { return (u64)(((u128)a * b) % GLP); }
static inline u64 g_add(u64 a, u64 b) { return (u64)(((u128)a + b) % GLP); }
static inline u64 g_sub(u64 a, u64 b) { return a >= b ? a - b : a + GLP - b; }
static u64 g_pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = g_mul(r, a);
        a = g_mul(a, a);
        e >>= 1;
    }
    return r;
}
static inline u64 g_inv(u64 a) { return g_pow(a, GLP - 2); }

// ---- small big integers on u32 limbs (little-endian), at most BIG_MAX limbs
#define BIG_MAX 40
struct Big {
    u32 v[BIG_MAX];
    int n;  // used limbs (may include leading zeros)
};
static Big big_zero() {
    Big r;
    memset(r.v, 0, sizeof(r.v));
    r.n = 0;
    return r;
}
static Big big_from(const u64 *limbs, int n) {
    Big r = big_zero();
    for (int i = 0; i < n; i++) r.v[i] = (u32)limbs[i];
    r.n = n;
    return r;
}
static void big_trim(Big &a) {
    while (a.n > 0 && a.v[a.n - 1] == 0) a.n--;
}
static int big_cmp(const Big &a, const Big &b) {
    for (int i = BIG_MAX - 1; i >= 0; i--)
        if (a.v[i] != b.v[i]) return a.v[i] < b.v[i] ? -1 : 1;
    return 0;
}
static Big big_add(const Big &a, const Big &b) {
    Big r = big_zero();
    u64 c = 0;
    int n = a.n > b.n ? a.n : b.n;
    for (int i = 0; i < n || c; i++) {
        u64 s = (u64)a.v[i] + b.v[i] + c;
        r.v[i] = (u32)s;
        c = s >> 32;
        r.n = i + 1;
    }
    if (r.n < n) r.n = n;
    return r;
}
static Big big_sub(const Big &a, const Big &b) {  // a >= b
    Big r = big_zero();
    int64_t c = 0;
    for (int i = 0; i < BIG_MAX; i++) {
        int64_t s = (int64_t)a.v[i] - b.v[i] + c;
        r.v[i] = (u32)s;
        c = s >> 32;
    }
    r.n = a.n;
    big_trim(r);
    return r;
}
static Big big_mul(const Big &a, const Big &b) {
    Big r = big_zero();
    for (int i = 0; i < a.n; i++) {
        u64 c = 0;
        for (int j = 0; j < b.n || c; j++) {
            u64 s = (u64)a.v[i] * (j < b.n ? b.v[j] : 0) + r.v[i + j] + c;
            r.v[i + j] = (u32)s;
            c = s >> 32;
        }
    }
    r.n = a.n + b.n;
    big_trim(r);
    return r;
}
// Knuth algorithm D.  q = a / b, r = a % b (b != 0)
static void big_divmod(Big a, Big b, Big &q, Big &r) {
    big_trim(a);
    big_trim(b);
    q = big_zero();
    if (big_cmp(a, b) < 0) {
        r = a;
        return;
    }
    if (b.n == 1) {
        u64 rem = 0;
        for (int i = a.n - 1; i >= 0; i--) {
            u64 cur = (rem << 32) | a.v[i];
            q.v[i] = (u32)(cur / b.v[0]);
            rem = cur % b.v[0];
        }
        q.n = a.n;
        big_trim(q);
        r = big_zero();
        r.v[0] = (u32)rem;
        r.n = rem ? 1 : 0;
        return;
    }
    int s = __builtin_clz(b.v[b.n - 1]);
    u32 un[BIG_MAX + 1], vn[BIG_MAX];
    int n = b.n, m = a.n - b.n;
    for (int i = n - 1; i > 0; i--) vn[i] = s ? (b.v[i] << s) | (b.v[i - 1] >> (32 - s)) : b.v[i];
    vn[0] = b.v[0] << s;
    un[a.n] = s ? a.v[a.n - 1] >> (32 - s) : 0;
    for (int i = a.n - 1; i > 0; i--) un[i] = s ? (a.v[i] << s) | (a.v[i - 1] >> (32 - s)) : a.v[i];
    un[0] = a.v[0] << s;
    for (int j = m; j >= 0; j--) {
        u64 num = ((u64)un[j + n] << 32) | un[j + n - 1];
        u64 qhat = num / vn[n - 1], rhat = num % vn[n - 1];
        while (qhat >= (1ULL << 32) || qhat * vn[n - 2] > ((rhat << 32) | un[j + n - 2])) {
            qhat--;
            rhat += vn[n - 1];
            if (rhat >= (1ULL << 32)) break;
        }
        int64_t borrow = 0;
        u64 carry = 0;
        for (int i = 0; i < n; i++) {
            u64 p = qhat * vn[i] + carry;
            carry = p >> 32;
            int64_t t = (int64_t)un[i + j] - (int64_t)(u32)p + borrow;
            un[i + j] = (u32)t;
            borrow = t >> 32;
        }
        int64_t t = (int64_t)un[j + n] - (int64_t)carry + borrow;
        un[j + n] = (u32)t;
        if (t < 0) {
            qhat--;
            u64 c = 0;
            for (int i = 0; i < n; i++) {
                u64 sum = (u64)un[i + j] + vn[i] + c;
                un[i + j] = (u32)sum;
                c = sum >> 32;
            }
            un[j + n] += (u32)c;
        }
        q.v[j] = (u32)qhat;
    }
    q.n = m + 1;
    big_trim(q);
    r = big_zero();
    for (int i = 0; i < n; i++) r.v[i] = s ? (un[i] >> s) | ((u64)un[i + 1] << (32 - s)) : un[i];
    r.n = n;
    big_trim(r);
}
static Big big_mod(const Big &a, const Big &m) {
    Big q, r;
    big_divmod(a, m, q, r);
    return r;
}
// a * b mod 2^255 - 19 for a, b < 2^256 (8 limbs): 2^256 = 38, fold twice, then subtract p while >= p
static bool is_p25519(const Big &m) {
    if (m.v[0] != 0xFFFFFFEDu || m.v[7] != 0x7FFFFFFFu) return false;
    for (int i = 1; i < 7; i++)
        if (m.v[i] != 0xFFFFFFFFu) return false;
    for (int i = 8; i < BIG_MAX; i++)
        if (m.v[i]) return false;
    return true;
}
static Big mulmod_25519(const Big &a, const Big &b, const Big &m) {
    u64 t[17] = {0};
    for (int i = 0; i < 8; i++) {
        u64 c = 0;
        for (int j = 0; j < 8; j++) {
            u64 s = (u64)a.v[i] * b.v[j] + t[i + j] + c;
            t[i + j] = (u32)s;
            c = s >> 32;
        }
        t[i + 8] = c;
    }
    // lo + 38 * hi
    u64 r[9], c = 0;
    for (int i = 0; i < 8; i++) {
        u64 s = t[i] + 38 * t[i + 8] + c;
        r[i] = (u32)s;
        c = s >> 32;
    }
    // c < 39: fold c * 2^256 = 38 c, and the top bit (2^255 = 19)
    u64 top = (r[7] >> 31) & 1;
    r[7] &= 0x7FFFFFFF;
    u64 add = 38 * c + 19 * top;
    for (int i = 0; i < 8 && add; i++) {
        u64 s = r[i] + add;
        r[i] = (u32)s;
        add = s >> 32;
    }
    Big out = big_zero();
    for (int i = 0; i < 8; i++) out.v[i] = (u32)r[i];
    out.n = 8;
    while (big_cmp(out, m) >= 0) out = big_sub(out, m);
    out.n = 8;
    big_trim(out);
    return out;
}
static Big big_mulmod(const Big &a, const Big &b, const Big &m) {
    if (a.n <= 8 && b.n <= 8 && is_p25519(m)) return mulmod_25519(a, b, m);
    return big_mod(big_mul(a, b), m);
}
static Big big_powmod(Big a, Big e, const Big &m) {
    Big r = big_zero();
    r.v[0] = 1;
    r.n = 1;
    big_trim(e);
    for (int i = 0; i < e.n * 32; i++) {
        if ((e.v[i >> 5] >> (i & 31)) & 1) r = big_mulmod(r, a, m);
        a = big_mulmod(a, a, m);
    }
    return r;
}
static Big big_small(u32 x) {
    Big r = big_zero();
    r.v[0] = x;
    r.n = x ? 1 : 0;
    return r;
}

enum {
    OP_CONST = 0, OP_ARITH, OP_SPLIT, OP_LE_SUM, OP_U32_MULADD, OP_ADD_MANY, OP_SUB_U32, OP_RANGE_CHECK, OP_COMPARISON, OP_IS_EQUAL,
    OP_RANDOM_ACCESS, OP_NN_ADD, OP_NN_SUB, OP_NN_MUL, OP_NN_INV, OP_DIV_REM, OP_DECOMPRESS, OP_POSEIDON,
    // gadgets of the in-circuit verifier (plonky2/recursion.py)
    OP_EXT_ARITH, OP_EXT_MUL, OP_EXT_INV, OP_EXPONENTIATION, OP_COSET_INTERP, OP_POSEIDON_MDS, OP_REDUCING, OP_REDUCING_EXT,
    // crypto/plonky2_u32/src/gates/{interleave_u32,uninterleave_to_u32,uninterleave_to_b32}.rs generators
    OP_INTERLEAVE, OP_UNINTERLEAVE
};

// quadratic extension GF(p)[X]/(X^2 - 7)
struct E2 {
    u64 a, b;
};
static inline E2 e_add(E2 x, E2 y) { return {g_add(x.a, y.a), g_add(x.b, y.b)}; }
static inline E2 e_sub(E2 x, E2 y) { return {g_sub(x.a, y.a), g_sub(x.b, y.b)}; }
static inline E2 e_mul(E2 x, E2 y) {
    return {g_add(g_mul(x.a, y.a), g_mul(7, g_mul(x.b, y.b))), g_add(g_mul(x.a, y.b), g_mul(x.b, y.a))};
}
static inline E2 e_scalar(u64 c, E2 x) { return {g_mul(c, x.a), g_mul(c, x.b)}; }
static inline E2 e_inv(E2 x) {
    u64 d = g_inv(g_sub(g_mul(x.a, x.a), g_mul(7, g_mul(x.b, x.b))));
    return {g_mul(x.a, d), g_mul(g_sub(0, x.b), d)};
}

struct Runner {
    std::vector<u64> val;
    std::vector<u32> epoch;
    u32 cur = 0;
    char err[200];
    bool failed = false;

    bool set(u32 slot, u64 v, u64 pc) {
        if (epoch[slot] == cur) {
            if (val[slot] != v) {
                snprintf(err, sizeof(err), "copy constraint violated at instruction %llu (slot %u: %llu != %llu)", (unsigned long long)pc,
                         slot, (unsigned long long)val[slot], (unsigned long long)v);
                failed = true;
                return false;
            }
            return true;
        }
        epoch[slot] = cur;
        val[slot] = v;
        return true;
    }
    bool fail(const char *what, u64 pc) {
        snprintf(err, sizeof(err), "%s at instruction %llu", what, (unsigned long long)pc);
        failed = true;
        return false;
    }

    // code (u32 words): [opcode, n_params, n_in, n_out, ins..., outs...] repeated; the parameters of all instructions are
    // consecutive in `params` (64-bit: constants are field elements)
    bool run(const u32 *code, u64 code_len, const int64_t *params, const u32 *in_slots, const u64 *in_vals, u32 n_inputs) {
        cur++;
        failed = false;
        for (u32 i = 0; i < n_inputs; i++)
            if (!set(in_slots[i], in_vals[i] % GLP, 0)) return false;
        std::vector<u64> in, out;
        u64 pc = 0, ip = 0, pp = 0;
        while (ip < code_len) {
            int op = (int)code[ip];
            u32 np = code[ip + 1], ni = code[ip + 2], no = code[ip + 3];
            const int64_t *pr = params + pp;
            const u32 *is = code + ip + 4, *os = is + ni;
            ip += 4 + ni + no;
            pp += np;
            pc++;
            in.resize(ni);
            for (u32 i = 0; i < ni; i++) {
                if (epoch[is[i]] != cur) return fail("input not available", pc);
                in[i] = val[is[i]];
            }
            out.clear();
            switch (op) {
                case OP_CONST: out.push_back((u64)pr[0]); break;
                case OP_ARITH: out.push_back(g_add(g_mul((u64)pr[0], g_mul(in[0], in[1])), g_mul((u64)pr[1], in[2]))); break;
                case OP_SPLIT: {
                    u64 base = (u64)pr[0], x = in[0];
                    for (u32 i = 0; i < (u32)pr[1]; i++) {
                        out.push_back(x % base);
                        x /= base;
                    }
                    if (x) return fail("split: value does not fit", pc);
                    break;
                }
                case OP_LE_SUM: {
                    u64 s = 0;
                    for (u32 i = ni; i-- > 0;) s = g_add(g_add(s, s), in[i]);
                    out.push_back(s);
                    break;
                }
                case OP_U32_MULADD: {
                    u128 o = (u128)in[0] * in[1] + in[2];
                    if (o >= GLP) return fail("u32 mul-add overflows the field", pc);
                    u64 v = (u64)o, lo = v & 0xFFFFFFFFULL, hi = v >> 32;
                    u64 diff = g_sub(0xFFFFFFFFULL, hi);
                    out.push_back(lo);
                    out.push_back(hi);
                    out.push_back(diff ? g_inv(diff) : 0);
                    for (int j = 0; j < 32; j++) out.push_back((v >> (2 * j)) & 3);
                    break;
                }
                case OP_ADD_MANY: {
                    u64 s = 0;
                    for (u32 i = 0; i < ni; i++) s += in[i];
                    u64 lo = s & 0xFFFFFFFFULL, hi = s >> 32;
                    if (hi >= 16) return fail("add-many carry does not fit", pc);
                    out.push_back(lo);
                    out.push_back(hi);
                    for (int j = 0; j < 16; j++) out.push_back((lo >> (2 * j)) & 3);
                    for (int j = 0; j < 2; j++) out.push_back((hi >> (2 * j)) & 3);
                    break;
                }
                case OP_SUB_U32: {
                    int64_t d = (int64_t)in[0] - (int64_t)in[1] - (int64_t)in[2];
                    u64 bout = d < 0;
                    int64_t res = d + ((int64_t)bout << 32);
                    if (res < 0 || res >= (1LL << 32)) return fail("u32 subtraction out of range", pc);
                    out.push_back((u64)res);
                    out.push_back(bout);
                    for (int j = 0; j < 16; j++) out.push_back(((u64)res >> (2 * j)) & 3);
                    break;
                }
                case OP_RANGE_CHECK:
                    for (u32 i = 0; i < ni; i++) {
                        if (in[i] >> 32) return fail("range check: value exceeds 32 bits", pc);
                        for (int j = 0; j < 16; j++) out.push_back((in[i] >> (2 * j)) & 3);
                    }
                    break;
                case OP_COMPARISON: {
                    u32 nc = (u32)pr[0], cb = (u32)pr[1];
                    u64 size = 1ULL << cb, msd = 0;
                    for (u32 i = 0; i < nc; i++) {
                        u64 ca = (in[0] >> (cb * i)) & (size - 1), cy = (in[1] >> (cb * i)) & (size - 1);
                        u64 diff = g_sub(cy, ca), eq = ca == cy;
                        out.push_back(ca);
                        out.push_back(cy);
                        out.push_back(eq ? 1 : g_inv(diff));
                        out.push_back(eq);
                        u64 inter = eq ? msd : 0;
                        out.push_back(inter);
                        msd = eq ? inter : g_add(inter, diff);
                    }
                    out.push_back(msd);
                    u64 top = g_add(size, msd);
                    if (top >= 2 * size) return fail("comparison: most significant difference out of range", pc);
                    for (u32 i = 0; i <= cb; i++) out.push_back((top >> i) & 1);
                    out.push_back((top >> cb) & 1);
                    break;
                }
                case OP_IS_EQUAL:
                    out.push_back(in[0] == in[1]);
                    out.push_back(in[0] == in[1] ? 0 : g_inv(g_sub(in[0], in[1])));
                    break;
                case OP_RANDOM_ACCESS: {
                    u32 bits = (u32)pr[0];
                    if (in[0] >> bits) return fail("random access: index out of range", pc);
                    out.push_back(in[1 + in[0]]);
                    for (u32 i = 0; i < bits; i++) out.push_back((in[0] >> i) & 1);
                    break;
                }
                case OP_NN_ADD:
                case OP_NN_SUB: {
                    u32 na = (u32)pr[0];
                    u64 ml[8];
                    for (int i = 0; i < 8; i++) ml[i] = (u64)pr[1 + i];
                    Big m = big_from(ml, 8);
                    Big a = big_mod(big_from(in.data(), (int)na), m), b = big_mod(big_from(in.data() + na, (int)(ni - na)), m);
                    Big r;
                    u64 ov;
                    if (op == OP_NN_ADD) {
                        Big t = big_add(a, b);
                        ov = big_cmp(t, m) > 0;      // nonnative.rs:487: strictly greater
                        r = ov ? big_sub(t, m) : t;
                    } else {
                        ov = big_cmp(a, b) < 0;
                        r = ov ? big_sub(big_add(a, m), b) : big_sub(a, b);
                    }
                    for (int i = 0; i < 8; i++) out.push_back(r.v[i]);
                    out.push_back(ov);
                    break;
                }
                case OP_NN_MUL: {
                    u32 na = (u32)pr[0], nover = (u32)pr[1];
                    u64 ml[8];
                    for (int i = 0; i < 8; i++) ml[i] = (u64)pr[2 + i];
                    Big m = big_from(ml, 8);
                    Big a = big_mod(big_from(in.data(), (int)na), m), b = big_mod(big_from(in.data() + na, (int)(ni - na)), m);
                    Big q, r;
                    big_divmod(big_mul(a, b), m, q, r);
                    for (int i = 0; i < 8; i++) out.push_back(r.v[i]);
                    for (u32 i = 0; i < nover; i++) out.push_back(q.v[i]);
                    break;
                }
                case OP_NN_INV: {
                    u32 n = (u32)pr[0];
                    u64 ml[8];
                    for (int i = 0; i < 8; i++) ml[i] = (u64)pr[1 + i];
                    Big m = big_from(ml, 8);
                    Big x = big_mod(big_from(in.data(), (int)ni), m);
                    Big two = big_zero();
                    two.v[0] = 2;
                    two.n = 1;
                    Big iv = big_powmod(x, big_sub(m, two), m);
                    Big one = big_zero();
                    one.v[0] = 1;
                    one.n = 1;
                    Big prod = big_mul(x, iv), q, r;
                    if (prod.n == 0) return fail("inverse of zero", pc);
                    big_divmod(big_sub(prod, one), m, q, r);
                    for (u32 i = 0; i < n; i++) out.push_back(iv.v[i]);
                    for (u32 i = 0; i < n; i++) out.push_back(q.v[i]);
                    break;
                }
                case OP_DIV_REM: {
                    u32 a_len = (u32)pr[0], n_div = (u32)pr[1], n_rem = (u32)pr[2];
                    Big a = big_from(in.data(), (int)a_len), b = big_from(in.data() + a_len, (int)(ni - a_len));
                    Big bt = b;
                    big_trim(bt);
                    if (bt.n == 0) return fail("division by zero", pc);
                    Big q, r;
                    big_divmod(a, b, q, r);
                    for (u32 i = 0; i < n_div; i++) out.push_back(q.v[i]);
                    for (u32 i = 0; i < n_rem; i++) out.push_back(r.v[i]);
                    break;
                }
                case OP_DECOMPRESS: {
                    // 256 bits, most significant first: sign of x, then y (curve25519 point decompression)
                    Big val = big_zero();
                    for (u32 i = 0; i < 256; i++)
                        if (in[i]) val.v[(255 - i) >> 5] |= 1u << ((255 - i) & 31);
                    val.n = 8;
                    u32 sign = val.v[7] >> 31;
                    val.v[7] &= 0x7FFFFFFF;
                    Big p = big_zero();
                    for (int i = 0; i < 8; i++) p.v[i] = 0xFFFFFFFFu;
                    p.v[0] = 0xFFFFFFEDu;
                    p.v[7] = 0x7FFFFFFFu;
                    p.n = 8;
                    Big one = big_zero();
                    one.v[0] = 1;
                    one.n = 1;
                    // d = -121665 / 121666 mod p
                    static const u32 DW[8] = {0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu};
                    Big d = big_zero();
                    memcpy(d.v, DW, 32);
                    d.n = 8;
                    Big y = big_mod(val, p);
                    Big yy = big_mulmod(y, y, p);
                    Big u = big_sub(big_add(yy, p), one);                     // y^2 - 1
                    u = big_mod(u, p);
                    Big v = big_mod(big_add(big_mulmod(d, yy, p), one), p);   // d y^2 + 1
                    Big two = big_zero();
                    two.v[0] = 2;
                    two.n = 1;
                    Big xx = big_mulmod(u, big_powmod(v, big_sub(p, two), p), p);
                    // x = xx^((p+3)/8); fix with sqrt(-1) = 2^((p-1)/4)
                    Big e = big_add(p, big_small(3)), eq, er;
                    big_divmod(e, big_small(8), eq, er);
                    Big x = big_powmod(xx, eq, p);
                    if (big_cmp(big_mulmod(x, x, p), xx) != 0) {
                        Big e2, e2r;
                        big_divmod(big_sub(p, one), big_small(4), e2, e2r);
                        x = big_mulmod(x, big_powmod(two, e2, p), p);
                    }
                    if (big_cmp(big_mulmod(x, x, p), xx) != 0) return fail("point decompression: not a curve point", pc);
                    if ((x.v[0] & 1) != sign) x = big_mod(big_sub(p, x), p);
                    for (int i = 0; i < 8; i++) out.push_back(x.v[i]);
                    for (int i = 0; i < 8; i++) out.push_back(val.v[i]);
                    break;
                }
                case OP_POSEIDON: {
                    u64 rows[135];
                    if (ni != 13 || in[12] > 1) return fail("poseidon: 12 inputs and a boolean swap expected", pc);
                    if (zklc_poseidon_gl_gate_rows(in.data(), in.data() + 12, 1, rows)) return fail("poseidon rows", pc);
                    for (int c = 12; c < 135; c++)
                        if (c != 24) out.push_back(rows[c]);
                    break;
                }
                case OP_EXT_ARITH: {   // params c0, c1; in m0, m1, addend
                    E2 o = e_add(e_scalar((u64)pr[0], e_mul({in[0], in[1]}, {in[2], in[3]})), e_scalar((u64)pr[1], {in[4], in[5]}));
                    out.push_back(o.a);
                    out.push_back(o.b);
                    break;
                }
                case OP_EXT_MUL: {
                    E2 o = e_scalar((u64)pr[0], e_mul({in[0], in[1]}, {in[2], in[3]}));
                    out.push_back(o.a);
                    out.push_back(o.b);
                    break;
                }
                case OP_EXT_INV: {
                    if (!in[0] && !in[1]) return fail("inverse of zero", pc);
                    E2 o = e_inv({in[0], in[1]});
                    out.push_back(o.a);
                    out.push_back(o.b);
                    break;
                }
                case OP_EXPONENTIATION: {   // in: base, n bits (little-endian); out: n intermediates, output
                    u32 n = ni - 1;
                    u64 cur = 1;
                    for (u32 i = 0; i < n; i++) {
                        u64 prev = i == 0 ? 1 : g_mul(cur, cur);
                        cur = in[1 + n - 1 - i] ? g_mul(prev, in[0]) : prev;
                        out.push_back(cur);
                    }
                    out.push_back(cur);
                    break;
                }
                case OP_COSET_INTERP: {   // params: subgroup_bits, degree, weights[2^bits]; in: shift, values, point
                    u32 sb = (u32)pr[0], d = (u32)pr[1], np_ = 1u << sb, nint = (np_ - 2) / (d - 1);
                    if (ni != 1 + 2 * np_ + 2 || np > 2 + 64 || np != 2 + np_) return fail("coset interpolation: bad arity", pc);
                    if (!in[0]) return fail("coset interpolation: zero shift", pc);
                    u64 gen = g_pow(1753635133440165772ULL, 1ULL << (32 - sb)), dom[64], x = 1;
                    for (u32 i = 0; i < np_; i++) {
                        dom[i] = x;
                        x = g_mul(x, gen);
                    }
                    E2 pt = {in[1 + 2 * np_], in[2 + 2 * np_]};
                    E2 shifted = e_scalar(g_inv(in[0]), pt);
                    out.push_back(shifted.a);
                    out.push_back(shifted.b);
                    E2 ev = {0, 0}, prod = {1, 0};
                    auto partial = [&](u32 s, u32 e) {
                        for (u32 i = s; i < e; i++) {
                            E2 term = e_sub(shifted, {dom[i], 0});
                            E2 wv = e_scalar((u64)pr[2 + i], {in[1 + 2 * i], in[2 + 2 * i]});
                            ev = e_add(e_mul(ev, term), e_mul(wv, prod));
                            prod = e_mul(prod, term);
                        }
                    };
                    partial(0, d);
                    for (u32 i = 0; i < nint; i++) {
                        out.push_back(ev.a);
                        out.push_back(ev.b);
                        out.push_back(prod.a);
                        out.push_back(prod.b);
                        u32 s = 1 + (d - 1) * (i + 1), e = s + d - 1 < np_ ? s + d - 1 : np_;
                        partial(s, e);
                    }
                    out.push_back(ev.a);
                    out.push_back(ev.b);
                    break;
                }
                case OP_POSEIDON_MDS: {   // 12 extension elements in, 12 out
                    static const u64 circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
                    for (int r = 0; r < 12; r++) {
                        u128 a = 0, b = 0;
                        for (int i = 0; i < 12; i++) {
                            int j = (i + r) % 12;
                            a += (u128)in[2 * j] * circ[i];
                            b += (u128)in[2 * j + 1] * circ[i];
                        }
                        if (r == 0) {
                            a += (u128)in[0] * 8;
                            b += (u128)in[1] * 8;
                        }
                        out.push_back((u64)(a % GLP));
                        out.push_back((u64)(b % GLP));
                    }
                    break;
                }
                case OP_REDUCING:
                case OP_REDUCING_EXT: {   // params n; in: alpha, old acc, n coefficients; out: the n accumulators (last = output)
                    u32 n = (u32)pr[0];
                    bool ext = op == OP_REDUCING_EXT;
                    if (ni != 4 + (ext ? 2 * n : n)) return fail("reducing: bad arity", pc);
                    E2 alpha = {in[0], in[1]}, acc = {in[2], in[3]};
                    for (u32 i = 0; i < n; i++) {
                        E2 c = ext ? E2{in[4 + 2 * i], in[5 + 2 * i]} : E2{in[4 + i], 0};
                        acc = e_add(e_mul(acc, alpha), c);
                        out.push_back(acc.a);
                        out.push_back(acc.b);
                    }
                    break;
                }
                case OP_INTERLEAVE: {   // in x (u32); out: x with its bits spread to the even positions, then 32 big-endian bits
                    if (in[0] >> 32) return fail("interleave: value exceeds 32 bits", pc);
                    u64 xi = 0;
                    for (int j = 0; j < 32; j++) xi |= ((in[0] >> j) & 1) << (2 * j);
                    out.push_back(xi);
                    for (int j = 0; j < 32; j++) out.push_back((in[0] >> (31 - j)) & 1);
                    break;
                }
                case OP_UNINTERLEAVE: {   // param to_b32; in x; out: evens, odds, 64 big-endian bits
                    const u32 step = pr[0] ? 2 : 1;
                    u64 ev = 0, od = 0;
                    for (int j = 0; j < 32; j++) {
                        ev |= ((in[0] >> (2 * j + 1)) & 1) << (step * j);
                        od |= ((in[0] >> (2 * j)) & 1) << (step * j);
                    }
                    out.push_back(ev);
                    out.push_back(od);
                    for (int j = 0; j < 64; j++) out.push_back((in[0] >> (63 - j)) & 1);
                    break;
                }
                default: return fail("unknown opcode", pc);
            }
            if (out.size() != no) return fail("output count mismatch", pc);
            for (u32 i = 0; i < no; i++)
                if (!set(os[i], out[i] % GLP, pc)) return false;
        }
        return true;
    }
};

// pool of interpreter states (value + epoch arrays are hundreds of MB for the Ed25519 circuit: keep them across calls)
#include <mutex>
static std::mutex g_pool_mutex;
static std::vector<Runner *> g_pool;
static Runner *runner_acquire(u32 n_slots) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i]->val.size() == n_slots) {
                Runner *r = g_pool[i];
                g_pool.erase(g_pool.begin() + i);
                return r;
            }
    }
    Runner *r = new Runner();
    r->val.assign(n_slots, 0);
    r->epoch.assign(n_slots, 0);
    return r;
}
static void runner_release(Runner *r) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (g_pool.size() < 16)
        g_pool.push_back(r);
    else
        delete r;
}
// frees the cached interpreter states (hundreds of MB each for the Ed25519 circuit)
extern "C" void zklc_plonky2_witness_release(void) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    for (Runner *r : g_pool) delete r;
    g_pool.clear();
}

// Runs the program for `n_witnesses` partial witnesses (input_values: n_witnesses x n_inputs) on up to `threads` host
// threads.  wire_slot / wire_index: the wire cells of every copy class (n_wire_entries), index = col * n_rows + row;
// wires_out: n_witnesses matrices of num_wires x n_rows u64 -- only the listed cells are written (the caller zero-fills the
// buffer once; the set of written cells is the same for every witness of a circuit).  pi_out: n_witnesses x n_pi.
// status[i] = 0 ok / 1 failed; err_out (optional): n_witnesses x 200 bytes of messages.
extern "C" int32_t zklc_plonky2_witness_run(const uint32_t *code, uint64_t code_len, const int64_t *params, uint32_t n_slots,
                                            const uint32_t *input_slots, uint32_t n_inputs, const uint64_t *input_values,
                                            uint32_t n_witnesses, const uint32_t *wire_slot, const uint32_t *wire_index,
                                            uint64_t n_wire_entries, uint32_t num_wires, uint32_t n_rows, uint64_t *wires_out,
                                            const uint32_t *pi_slots, uint32_t n_pi, uint64_t *pi_out, int32_t *status, char *err_out,
                                            uint32_t threads) {
    if (!code || !status || (n_inputs && (!input_slots || !input_values)) || !wires_out) return -1;
    if (threads == 0) threads = 1;
    if (threads > n_witnesses) threads = n_witnesses ? n_witnesses : 1;
    std::atomic<u32> next(0);
    auto worker = [&]() {
        Runner *rp = runner_acquire(n_slots);
        Runner &r = *rp;
        for (;;) {
            u32 w = next.fetch_add(1);
            if (w >= n_witnesses) break;
            bool ok = r.run(code, code_len, params, input_slots, input_values + (size_t)w * n_inputs, n_inputs);
            u64 *wires = wires_out + (size_t)w * num_wires * n_rows;
            if (ok) {
                for (u64 k = 0; k < n_wire_entries; k++) {
                    u32 s = wire_slot[k];
                    if (r.epoch[s] != r.cur) continue;  // unconstrained cell of a class nobody assigned: stays as it is (zero)
                    wires[wire_index[k]] = r.val[s];
                }
                for (u32 k = 0; k < n_pi; k++) {
                    if (r.epoch[pi_slots[k]] != r.cur) {
                        ok = false;
                        snprintf(r.err, sizeof(r.err), "public input %u was never assigned", k);
                        break;
                    }
                    pi_out[(size_t)w * n_pi + k] = r.val[pi_slots[k]];
                }
            }
            status[w] = ok ? 0 : 1;
            if (err_out) {
                if (ok)
                    err_out[(size_t)w * 200] = 0;
                else
                    memcpy(err_out + (size_t)w * 200, r.err, 200);
            }
        }
        runner_release(rp);
    };
    std::vector<std::thread> pool;
    for (u32 t = 1; t < threads; t++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return 0;
}
