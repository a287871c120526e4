// Semantics of the witness-generator instructions, shared by the host interpreter (plonky2_witness.cpp, plain C++) and the
// device interpreter (plonky2_witness_dev.hip, one lane per (instruction, witness)).  One instruction = one generator of the
// reference:
//   crypto/plonky2_ed25519/src/gadgets/nonnative.rs:447-705  NonNative{Addition,Subtraction,Multiplication,Inverse}Generator
//   crypto/plonky2_ed25519/src/gadgets/curve.rs:327-370      CurvePointDecompressionGenerator
//   crypto/plonky2_ecdsa/src/gadgets/biguint.rs:417-470      BigUintDivRemGenerator
//   crypto/plonky2_u32/src/gates/*.rs `generators()`          u32 arithmetic / add-many / subtraction / range-check / comparison /
//                                                             interleave gate generators
//   plonky2 (un-vendored) ArithmeticGate / BaseSumGate / RandomAccessGate / PoseidonGate / EqualityGenerator and the gadgets
//   of the in-circuit verifier (extension arithmetic, exponentiation, coset interpolation, reducing gates).
// `wit_exec` reads its inputs through io.in(i), appends its outputs with io.out(v) (false = the slot already holds another
// value: the reference's "copy constraint violated") and reports a failure with io.fail(code).
#pragma once
#include <stdint.h>
#include <string.h>
#include "goldilocks.cuh"
#include "wit25519.cuh"

#if defined(__HIPCC__)
#define WIT_FN __host__ __device__ static
#else
#define WIT_FN static
#endif

enum {
    WIT_OK = 0, WIT_ERR_COPY, WIT_ERR_INPUT_NA, WIT_ERR_SPLIT, WIT_ERR_MULADD, WIT_ERR_ADD_MANY, WIT_ERR_SUB, WIT_ERR_RANGE,
    WIT_ERR_COMPARISON, WIT_ERR_RANDOM_ACCESS, WIT_ERR_INV_ZERO, WIT_ERR_DIV_ZERO, WIT_ERR_DECOMPRESS, WIT_ERR_POSEIDON,
    WIT_ERR_COSET_ARITY, WIT_ERR_COSET_SHIFT, WIT_ERR_REDUCING, WIT_ERR_INTERLEAVE, WIT_ERR_OPCODE, WIT_ERR_OUT_COUNT, WIT_ERR_PI, WIT_ERR_NEEDS_HEAVY,
    WIT_NUM_ERRORS
};
static inline const char *wit_strerror(int code) {
    static const char *const msg[WIT_NUM_ERRORS] = {
        "ok", "copy constraint violated", "input not available", "split: value does not fit", "u32 mul-add overflows the field",
        "add-many carry does not fit", "u32 subtraction out of range", "range check: value exceeds 32 bits",
        "comparison: most significant difference out of range", "random access: index out of range", "inverse of zero",
        "division by zero", "point decompression: not a curve point", "poseidon: 12 inputs and a boolean swap expected",
        "coset interpolation: bad arity", "coset interpolation: zero shift", "reducing: bad arity",
        "interleave: value exceeds 32 bits", "unknown opcode", "output count mismatch", "public input was never assigned",
        "instruction scheduled into the wrong kernel class"};
    return code >= 0 && code < WIT_NUM_ERRORS ? msg[code] : "?";
}

#if defined(__HIP_DEVICE_COMPILE__)
WIT_FN u64 g_mul(u64 a, u64 b) { return gl_mul(a, b); }
WIT_FN u64 g_add(u64 a, u64 b) { return gl_add(a, b); }
#else
typedef unsigned __int128 wit_u128;
WIT_FN u64 g_mul(u64 a, u64 b) { return (u64)(((wit_u128)a * b) % GL_P); }
WIT_FN u64 g_add(u64 a, u64 b) { return (u64)(((wit_u128)a + b) % GL_P); }
#endif
WIT_FN u64 g_sub(u64 a, u64 b) { return a >= b ? a - b : a + GL_P - b; }
WIT_FN u64 g_pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = g_mul(r, a);
        a = g_mul(a, a);
        e >>= 1;
    }
    return r;
}
// a^(p - 2) with p - 2 = (2^31 - 1) 2^33 + 2^32 - 1: 64 squarings and 9 multiplications (a plain square-and-multiply takes 127
// products; the comparison and u32 gates' generators invert on nearly every level of a witness program)
WIT_FN u64 g_sqn(u64 a, int n) {
    for (int i = 0; i < n; i++) a = g_mul(a, a);
    return a;
}
WIT_FN u64 g_inv(u64 a) {
    u64 t2 = g_mul(g_mul(a, a), a);                 // 2^2 - 1
    u64 t3 = g_mul(g_mul(t2, t2), a);               // 2^3 - 1
    u64 t6 = g_mul(g_sqn(t3, 3), t3);
    u64 t12 = g_mul(g_sqn(t6, 6), t6);
    u64 t24 = g_mul(g_sqn(t12, 12), t12);
    u64 t30 = g_mul(g_sqn(t24, 6), t6);
    u64 t31 = g_mul(g_mul(t30, t30), a);            // 2^31 - 1
    u64 t32 = g_mul(g_mul(t31, t31), a);            // 2^32 - 1
    return g_mul(g_sqn(t31, 33), t32);
}
// 1 / d for a small integer 0 < d <= 64: (p t + 1) / d with t the solution of p t = -1 (mod d); no field arithmetic
WIT_FN u64 g_inv_small(u64 d) {
    if (d <= 3) return d == 1 ? 1 : d == 2 ? 0x7FFFFFFF80000001ULL : 0xAAAAAAAA00000001ULL;   // the 2-bit chunks of the u32 comparisons
    u64 qd = GL_P / d, rd = GL_P % d, t = 0;
    while ((rd * t + 1) % d) t++;
    return qd * t + (rd * t + 1) / d;
}

// ---- small big integers on u32 limbs (little-endian), at most BIG_MAX limbs
#define BIG_MAX 40
struct Big {
    u32 v[BIG_MAX];
    int n;  // used limbs (may include leading zeros)
};
WIT_FN Big big_zero() {
    Big r;
    memset(r.v, 0, sizeof(r.v));
    r.n = 0;
    return r;
}
WIT_FN Big big_from(const u64 *limbs, int n) {
    Big r = big_zero();
    for (int i = 0; i < n; i++) r.v[i] = (u32)limbs[i];
    r.n = n;
    return r;
}
WIT_FN void big_trim(Big &a) {
    while (a.n > 0 && a.v[a.n - 1] == 0) a.n--;
}
WIT_FN int big_cmp(const Big &a, const Big &b) {
    for (int i = BIG_MAX - 1; i >= 0; i--)
        if (a.v[i] != b.v[i]) return a.v[i] < b.v[i] ? -1 : 1;
    return 0;
}
WIT_FN Big big_add(const Big &a, const Big &b) {
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
WIT_FN Big big_sub(const Big &a, const Big &b) {  // a >= b
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
WIT_FN Big big_mul(const Big &a, const Big &b) {
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
WIT_FN int wit_clz(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)x);
#else
    return __builtin_clz(x);
#endif
}
// Knuth algorithm D.  q = a / b, r = a % b (b != 0)
WIT_FN void big_divmod(Big a, Big b, Big &q, Big &r) {
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
    int s = wit_clz(b.v[b.n - 1]);
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
WIT_FN Big big_mod(const Big &a, const Big &m) {
    Big q, r;
    big_divmod(a, m, q, r);
    return r;
}
// a * b mod 2^255 - 19 for a, b < 2^256 (8 limbs): 2^256 = 38, fold twice, then subtract p while >= p
WIT_FN bool is_p25519(const Big &m) {
    if (m.v[0] != 0xFFFFFFEDu || m.v[7] != 0x7FFFFFFFu) return false;
    for (int i = 1; i < 7; i++)
        if (m.v[i] != 0xFFFFFFFFu) return false;
    for (int i = 8; i < BIG_MAX; i++)
        if (m.v[i]) return false;
    return true;
}
WIT_FN Big mulmod_25519(const Big &a, const Big &b, const Big &m) {
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
WIT_FN Big big_mulmod(const Big &a, const Big &b, const Big &m) {
    if (a.n <= 8 && b.n <= 8 && is_p25519(m)) return mulmod_25519(a, b, m);
    return big_mod(big_mul(a, b), m);
}
WIT_FN Big big_powmod(Big a, Big e, const Big &m) {
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
WIT_FN Big big_small(u32 x) {
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
WIT_FN E2 e_add(E2 x, E2 y) { return {g_add(x.a, y.a), g_add(x.b, y.b)}; }
WIT_FN E2 e_sub(E2 x, E2 y) { return {g_sub(x.a, y.a), g_sub(x.b, y.b)}; }
WIT_FN E2 e_mul(E2 x, E2 y) {
    return {g_add(g_mul(x.a, y.a), g_mul(7, g_mul(x.b, y.b))), g_add(g_mul(x.a, y.b), g_mul(x.b, y.a))};
}
WIT_FN E2 e_scalar(u64 c, E2 x) { return {g_mul(c, x.a), g_mul(c, x.b)}; }
WIT_FN E2 e_inv(E2 x) {
    u64 d = g_inv(g_sub(g_mul(x.a, x.a), g_mul(7, g_mul(x.b, x.b))));
    return {g_mul(x.a, d), g_mul(g_sub(0, x.b), d)};
}


WIT_FN void wit_mac128(u64 &lo, u64 &hi, u64 x, u64 c) {
    u64 pl, ph;
    gl_mul_wide(x, c, pl, ph);
    lo += pl;
    hi += ph + (lo < pl);
}

// eight words of a non-native operand (n <= 8 limbs), reduced mod 2^255 - 19
template <class IO>
WIT_FN void wit_w8_in(const IO &io, u32 first, u32 n, u32 *w) {
#pragma unroll
    for (u32 i = 0; i < 8; i++) w[i] = i < n ? (u32)io.in(first + i) : 0;
    w25519_reduce(w);
}

template <class IO>
WIT_FN Big wit_big_in(const IO &io, u32 first, int n) {
    Big r = big_zero();
    for (int i = 0; i < n; i++) r.v[i] = (u32)io.in(first + (u32)i);
    r.n = n;
    return r;
}

// PoseidonGate row (135 wires) from the 12 inputs and the swap flag: the outputs are the row's wires 12..134 without wire 24
// (the swap flag), in that order; they are produced out of order, so they go through io.out_at(index, value)
WIT_FN u32 wit_prow(u32 col) { return col - 12 - (col > 24 ? 1 : 0); }
#if defined(__HIP_DEVICE_COMPILE__)
#include "poseidon_gl.cuh"
template <class IO>
WIT_FN bool wit_poseidon_rows(IO &io) {
    u64 sw = io.in(12);
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 a = io.in(i), b = io.in(i + 4);
        u64 delta = sw ? gl_sub(b, a) : 0;
        if (!io.out_at(wit_prow(25 + i), delta)) return false;
        s[i] = gl_add(a, delta);
        s[i + 4] = gl_sub(b, delta);
    }
#pragma unroll
    for (int i = 8; i < 12; i++) s[i] = io.in(i);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], PGL_RC[12 * r + i]);
        if (r)
#pragma unroll
            for (int i = 0; i < 12; i++)
                if (!io.out_at(wit_prow(29 + 12 * (r - 1) + i), s[i])) return false;
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = pgl_sbox(s[i]);
        pgl_mds(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], PGL_FP_FIRST[i]);
    {
        u64 t[12];
        t[0] = s[0];
#pragma unroll 1
        for (int d = 1; d < 12; d++) {
            gl_acc160 a = {0, 0, 0};
#pragma unroll
            for (int r = 1; r < 12; r++) gl_acc_mul(a, s[r], PGL_FP_INIT[(r - 1) * 11 + d - 1]);
            u64 v = gl_acc_reduce(a);
            // t[d] with a loop-variable index: rotate into place (static register indices)
#pragma unroll
            for (int q = 1; q < 11; q++) t[q] = t[q + 1];
            t[11] = v;
        }
#pragma unroll
        for (int i = 1; i < 12; i++) s[i] = t[i];
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        if (!io.out_at(wit_prow(65 + r), s[0])) return false;
        u64 s0 = gl_add(pgl_sbox(s[0]), PGL_FP_RC[r]);
        gl_acc160 a = {0, 0, 0};
        gl_acc_mul(a, s0, 25);
#pragma unroll
        for (int j = 1; j < 12; j++) gl_acc_mul(a, s[j], PGL_FP_WHATS[r * 11 + j - 1]);
#pragma unroll
        for (int j = 1; j < 12; j++) s[j] = gl_add(s[j], gl_mul(s0, PGL_FP_VS[r * 11 + j - 1]));
        s[0] = gl_acc_reduce(a);
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            s[i] = gl_add(s[i], PGL_RC[12 * (26 + r) + i]);
            if (!io.out_at(wit_prow(87 + 12 * r + i), s[i])) return false;
            s[i] = pgl_sbox(s[i]);
        }
        pgl_mds(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++)
        if (!io.out_at(wit_prow(12 + i), s[i])) return false;
    return true;
}
#else
#include "plonky2_host.h"
template <class IO>
WIT_FN bool wit_poseidon_rows(IO &io) {
    u64 pin[12], sw = io.in(12), rows[135];
    for (int c = 0; c < 12; c++) pin[c] = io.in(c);
    if (zklc_poseidon_gl_gate_rows(pin, &sw, 1, rows)) return io.fail(WIT_ERR_POSEIDON);
    for (u32 c = 12; c < 135; c++)
        if (c != 24 && !io.out_at(wit_prow(c), rows[c])) return false;
    return true;
}
#endif

// instructions the device must run in the kernel with the generic big-integer code (wit_exec<true>)
WIT_FN bool wit_is_heavy(int op, const int64_t *pr, u32 ni);

#if defined(WIT_NO_FAST25519)
#define WIT_FAST 0
#else
#define WIT_FAST 1
#endif
#define WIT_OUT(v)                       \
    do {                                 \
        if (!io.out(v)) return false;    \
    } while (0)

// executes one instruction; pr: its parameters (np of them), ni / no: input / output counts
WIT_FN bool wit_is_heavy(int op, const int64_t *pr, u32 ni) {
    switch (op) {
        case OP_NN_ADD:
        case OP_NN_SUB: return !(WIT_FAST && w25519_is_p(pr + 1) && (u32)pr[0] <= 8 && ni - (u32)pr[0] <= 8);
        case OP_NN_MUL: return !(WIT_FAST && w25519_is_p(pr + 2) && (u32)pr[0] <= 8 && ni - (u32)pr[0] <= 8 && (u32)pr[1] <= 9);
        case OP_NN_INV: return !(WIT_FAST && w25519_is_p(pr + 1) && ni <= 8 && (u32)pr[0] <= 8);
        case OP_DIV_REM: return true;
        case OP_DECOMPRESS: return !WIT_FAST;
        default: return false;
    }
}

// HEAVY = false compiles the generic big-integer paths out (WIT_ERR_NEEDS_HEAVY instead): the device runs those instructions --
// a handful per circuit, see wit_is_heavy -- in a second kernel, so that the kernel of the common instructions needs no scratch
template <bool HEAVY, class IO>
WIT_FN bool wit_exec(int op, const int64_t *pr, u32 np, u32 ni, u32 no, IO &io) {
    (void)np;
    (void)no;
    switch (op) {
        case OP_CONST: WIT_OUT((u64)pr[0]); break;
        case OP_ARITH: WIT_OUT(g_add(g_mul((u64)pr[0], g_mul(io.in(0), io.in(1))), g_mul((u64)pr[1], io.in(2)))); break;
        case OP_SPLIT: {
            u64 base = (u64)pr[0], x = io.in(0);
            if ((base & (base - 1)) == 0) {                // bits / base-4 limbs: shifts (the device has no 64-bit divider)
                u32 sh = 0;
                while ((1ULL << sh) < base) sh++;
                for (u32 i = 0; i < (u32)pr[1]; i++) {
                    WIT_OUT(x & (base - 1));
                    x = sh < 64 ? x >> sh : 0;
                }
            } else
                for (u32 i = 0; i < (u32)pr[1]; i++) {
                    WIT_OUT(x % base);
                    x /= base;
                }
            if (x) return io.fail(WIT_ERR_SPLIT);
            break;
        }
        case OP_LE_SUM: {
            u64 s = 0;
            for (u32 i = ni; i-- > 0;) s = g_add(g_add(s, s), io.in(i));
            WIT_OUT(s);
            break;
        }
        case OP_U32_MULADD: {
            u64 plo, phi;
            gl_mul_wide(io.in(0), io.in(1), plo, phi);
            u64 v = plo + io.in(2);
            if (v < plo) phi++;
            if (phi || v >= GL_P) return io.fail(WIT_ERR_MULADD);
            u64 lo = v & 0xFFFFFFFFULL, hi = v >> 32;
            u64 diff = g_sub(0xFFFFFFFFULL, hi);
            WIT_OUT(lo);
            WIT_OUT(hi);
            WIT_OUT(diff ? g_inv(diff) : 0);
            for (int j = 0; j < 32; j++) WIT_OUT((v >> (2 * j)) & 3);
            break;
        }
        case OP_ADD_MANY: {
            u64 s = 0;
            for (u32 i = 0; i < ni; i++) s += io.in(i);
            u64 lo = s & 0xFFFFFFFFULL, hi = s >> 32;
            if (hi >= 16) return io.fail(WIT_ERR_ADD_MANY);
            WIT_OUT(lo);
            WIT_OUT(hi);
            for (int j = 0; j < 16; j++) WIT_OUT((lo >> (2 * j)) & 3);
            for (int j = 0; j < 2; j++) WIT_OUT((hi >> (2 * j)) & 3);
            break;
        }
        case OP_SUB_U32: {
            int64_t d = (int64_t)io.in(0) - (int64_t)io.in(1) - (int64_t)io.in(2);
            u64 bout = d < 0;
            int64_t res = d + ((int64_t)bout << 32);
            if (res < 0 || res >= (1LL << 32)) return io.fail(WIT_ERR_SUB);
            WIT_OUT((u64)res);
            WIT_OUT(bout);
            for (int j = 0; j < 16; j++) WIT_OUT(((u64)res >> (2 * j)) & 3);
            break;
        }
        case OP_RANGE_CHECK:
            for (u32 i = 0; i < ni; i++) {
                const u64 x = io.in(i);             // (a slot read is a memory round trip on the device: read once)
                if (x >> 32) return io.fail(WIT_ERR_RANGE);
                for (int j = 0; j < 16; j++) WIT_OUT((x >> (2 * j)) & 3);
            }
            break;
        case OP_COMPARISON: {
            u32 nc = (u32)pr[0], cb = (u32)pr[1];
            u64 size = 1ULL << cb, msd = 0;
            const u64 x0 = io.in(0), x1 = io.in(1);
            for (u32 i = 0; i < nc; i++) {
                u64 ca = (x0 >> (cb * i)) & (size - 1), cy = (x1 >> (cb * i)) & (size - 1);
                u64 diff = g_sub(cy, ca), eq = ca == cy;
                WIT_OUT(ca);
                WIT_OUT(cy);
                u64 ad = cy > ca ? cy - ca : ca - cy;      // |difference| < 2^chunk_bits: a division instead of an exponentiation
                u64 dinv = eq ? 1 : ad <= 64 ? (cy > ca ? g_inv_small(ad) : GL_P - g_inv_small(ad)) : g_inv(diff);
                WIT_OUT(dinv);
                WIT_OUT(eq);
                u64 inter = eq ? msd : 0;
                WIT_OUT(inter);
                msd = eq ? inter : g_add(inter, diff);
            }
            WIT_OUT(msd);
            u64 top = g_add(size, msd);
            if (top >= 2 * size) return io.fail(WIT_ERR_COMPARISON);
            for (u32 i = 0; i <= cb; i++) WIT_OUT((top >> i) & 1);
            WIT_OUT((top >> cb) & 1);
            break;
        }
        case OP_IS_EQUAL:
        {
            const u64 x0 = io.in(0), x1 = io.in(1);
            WIT_OUT(x0 == x1);
            WIT_OUT(x0 == x1 ? 0 : g_inv(g_sub(x0, x1)));
            break;
        }
        case OP_RANDOM_ACCESS: {
            u32 bits = (u32)pr[0];
            const u64 idx = io.in(0);
            if (idx >> bits) return io.fail(WIT_ERR_RANDOM_ACCESS);
            WIT_OUT(io.in(1 + (u32)idx));
            for (u32 i = 0; i < bits; i++) WIT_OUT((idx >> i) & 1);
            break;
        }
        case OP_NN_ADD:
        case OP_NN_SUB: {
            u32 na = (u32)pr[0];
            if (WIT_FAST && w25519_is_p(pr + 1) && na <= 8 && ni - na <= 8) {      // the Ed25519 base field: word arithmetic in registers
                u32 a[8], b[8], r[8];
                wit_w8_in(io, 0, na, a);
                wit_w8_in(io, na, ni - na, b);
                u64 ov;
                if (op == OP_NN_ADD) {
                    u32 t[8], t19[8];
                    u64 c = 0, c18 = 18, c19 = 19;
                    u32 top18 = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        c += (u64)a[i] + b[i];
                        t[i] = (u32)c;
                        c >>= 32;
                        c18 += t[i];
                        top18 = (u32)c18;
                        c18 >>= 32;
                        c19 += t[i];
                        t19[i] = (u32)c19;
                        c19 >>= 32;
                    }
                    ov = top18 >> 31;                                   // a + b > p  <=>  a + b + 18 >= 2^255  (a + b < 2^256)
                    t19[7] &= 0x7FFFFFFFu;                              // a + b - p = a + b + 19 - 2^255
#pragma unroll
                    for (int i = 0; i < 8; i++) r[i] = ov ? t19[i] : t[i];
                } else {
                    const u32 pw[8] = {0xFFFFFFEDu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu};
                    u32 d[8];
                    int64_t bw = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        int64_t t = (int64_t)a[i] - b[i] + bw;
                        d[i] = (u32)t;
                        bw = t >> 32;
                    }
                    ov = bw != 0;
                    u64 c = 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        c += (u64)d[i] + pw[i];
                        r[i] = ov ? (u32)c : d[i];
                        c >>= 32;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) WIT_OUT(r[i]);
                WIT_OUT(ov);
                break;
            }
            if constexpr (!HEAVY) {
                return io.fail(WIT_ERR_NEEDS_HEAVY);
            } else {
                u64 ml[8];
                for (int i = 0; i < 8; i++) ml[i] = (u64)pr[1 + i];
                Big m = big_from(ml, 8);
                Big a = big_mod(wit_big_in(io, 0, (int)na), m), b = big_mod(wit_big_in(io, na, (int)(ni - na)), m);
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
                for (int i = 0; i < 8; i++) WIT_OUT(r.v[i]);
                WIT_OUT(ov);
            }
            break;
        }
        case OP_NN_MUL: {
            u32 na = (u32)pr[0], nover = (u32)pr[1];
            if (WIT_FAST && w25519_is_p(pr + 2) && na <= 8 && ni - na <= 8 && nover <= 9) {
                u32 a[8], b[8], prod[16], q[9], r[8];
                wit_w8_in(io, 0, na, a);
                wit_w8_in(io, na, ni - na, b);
                w25519_mul_wide(a, b, prod);
                w25519_divmod(prod, q, r);
#pragma unroll
                for (int i = 0; i < 8; i++) WIT_OUT(r[i]);
                for (u32 i = 0; i < nover; i++) WIT_OUT(q[i]);
                break;
            }
            if constexpr (!HEAVY) {
                return io.fail(WIT_ERR_NEEDS_HEAVY);
            } else {
                u64 ml[8];
                for (int i = 0; i < 8; i++) ml[i] = (u64)pr[2 + i];
                Big m = big_from(ml, 8);
                Big a = big_mod(wit_big_in(io, 0, (int)na), m), b = big_mod(wit_big_in(io, na, (int)(ni - na)), m);
                Big q, r;
                big_divmod(big_mul(a, b), m, q, r);
                for (int i = 0; i < 8; i++) WIT_OUT(r.v[i]);
                for (u32 i = 0; i < nover; i++) WIT_OUT(q.v[i]);
            }
            break;
        }
        case OP_NN_INV: {
            u32 n = (u32)pr[0];
            if (WIT_FAST && w25519_is_p(pr + 1) && ni <= 8 && n <= 8) {
                u32 x[8], iv[8], prod[16], q[9], r[8];
                wit_w8_in(io, 0, ni, x);
                u32 any = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) any |= x[i];
                if (!any) return io.fail(WIT_ERR_INV_ZERO);
                w25519_inv(x, iv);
                w25519_mul_wide(x, iv, prod);
                w25519_divmod(prod, q, r);                              // x x^-1 = q p + 1
                for (u32 i = 0; i < n; i++) WIT_OUT(iv[i]);
                for (u32 i = 0; i < n; i++) WIT_OUT(q[i]);
                break;
            }
            if constexpr (!HEAVY) {
                return io.fail(WIT_ERR_NEEDS_HEAVY);
            } else {
                u64 ml[8];
                for (int i = 0; i < 8; i++) ml[i] = (u64)pr[1 + i];
                Big m = big_from(ml, 8);
                Big x = big_mod(wit_big_in(io, 0, (int)ni), m);
                Big two = big_zero();
                two.v[0] = 2;
                two.n = 1;
                Big iv = big_powmod(x, big_sub(m, two), m);
                Big one = big_zero();
                one.v[0] = 1;
                one.n = 1;
                Big prod = big_mul(x, iv), q, r;
                if (prod.n == 0) return io.fail(WIT_ERR_INV_ZERO);
                big_divmod(big_sub(prod, one), m, q, r);
                for (u32 i = 0; i < n; i++) WIT_OUT(iv.v[i]);
                for (u32 i = 0; i < n; i++) WIT_OUT(q.v[i]);
            }
            break;
        }
        case OP_DIV_REM: {
            u32 a_len = (u32)pr[0], n_div = (u32)pr[1], n_rem = (u32)pr[2];
            if constexpr (!HEAVY) {
                (void)a_len; (void)n_div; (void)n_rem;
                return io.fail(WIT_ERR_NEEDS_HEAVY);
            } else {
                Big a = wit_big_in(io, 0, (int)a_len), b = wit_big_in(io, a_len, (int)(ni - a_len));
                Big bt = b;
                big_trim(bt);
                if (bt.n == 0) return io.fail(WIT_ERR_DIV_ZERO);
                Big q, r;
                big_divmod(a, b, q, r);
                for (u32 i = 0; i < n_div; i++) WIT_OUT(q.v[i]);
                for (u32 i = 0; i < n_rem; i++) WIT_OUT(r.v[i]);
            }
            break;
        }
        case OP_DECOMPRESS: {
            // 256 bits, most significant first: sign of x, then y (curve25519 point decompression)
#if !defined(WIT_NO_FAST25519)
            {
                u32 yw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xw[8];
                for (u32 i = 0; i < 256; i++)
                    if (io.in(i)) yw[(255 - i) >> 5] |= 1u << ((255 - i) & 31);
                u32 sign = yw[7] >> 31;
                yw[7] &= 0x7FFFFFFFu;
                if (!w25519_decompress(yw, sign, xw)) return io.fail(WIT_ERR_DECOMPRESS);
                for (int i = 0; i < 8; i++) WIT_OUT(xw[i]);
                for (int i = 0; i < 8; i++) WIT_OUT(yw[i]);
                break;
            }
#endif
            if constexpr (!HEAVY) {
                return io.fail(WIT_ERR_NEEDS_HEAVY);
            } else {
                Big val = big_zero();
                for (u32 i = 0; i < 256; i++)
                    if (io.in(i)) val.v[(255 - i) >> 5] |= 1u << ((255 - i) & 31);
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
                const u32 DW[8] = {0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu};
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
                if (big_cmp(big_mulmod(x, x, p), xx) != 0) return io.fail(WIT_ERR_DECOMPRESS);
                if ((x.v[0] & 1) != sign) x = big_mod(big_sub(p, x), p);
                for (int i = 0; i < 8; i++) WIT_OUT(x.v[i]);
                for (int i = 0; i < 8; i++) WIT_OUT(val.v[i]);
            }
            break;
        }
        case OP_POSEIDON: {
            if (ni != 13 || io.in(12) > 1) return io.fail(WIT_ERR_POSEIDON);
            if (!wit_poseidon_rows(io)) return false;
            break;
        }
        case OP_EXT_ARITH: {   // params c0, c1; in m0, m1, addend
            E2 o = e_add(e_scalar((u64)pr[0], e_mul({io.in(0), io.in(1)}, {io.in(2), io.in(3)})), e_scalar((u64)pr[1], {io.in(4), io.in(5)}));
            WIT_OUT(o.a);
            WIT_OUT(o.b);
            break;
        }
        case OP_EXT_MUL: {
            E2 o = e_scalar((u64)pr[0], e_mul({io.in(0), io.in(1)}, {io.in(2), io.in(3)}));
            WIT_OUT(o.a);
            WIT_OUT(o.b);
            break;
        }
        case OP_EXT_INV: {
            if (!io.in(0) && !io.in(1)) return io.fail(WIT_ERR_INV_ZERO);
            E2 o = e_inv({io.in(0), io.in(1)});
            WIT_OUT(o.a);
            WIT_OUT(o.b);
            break;
        }
        case OP_EXPONENTIATION: {   // in: base, n bits (little-endian); out: n intermediates, output
            u32 n = ni - 1;
            u64 cur = 1;
            const u64 base = io.in(0);
            for (u32 i = 0; i < n; i++) {
                u64 prev = i == 0 ? 1 : g_mul(cur, cur);
                cur = io.in(1 + n - 1 - i) ? g_mul(prev, base) : prev;
                WIT_OUT(cur);
            }
            WIT_OUT(cur);
            break;
        }
        case OP_COSET_INTERP: {   // params: subgroup_bits, degree, weights[2^bits]; in: shift, values, point
            u32 sb = (u32)pr[0], d = (u32)pr[1], np_ = 1u << sb, nint = (np_ - 2) / (d - 1);
            if (ni != 1 + 2 * np_ + 2 || np > 2 + 64 || np != 2 + np_) return io.fail(WIT_ERR_COSET_ARITY);
            if (!io.in(0)) return io.fail(WIT_ERR_COSET_SHIFT);
            u64 gen = g_pow(1753635133440165772ULL, 1ULL << (32 - sb)), dom_i = 1;     // the subgroup point w^i, i = the next index
            E2 pt = {io.in(1 + 2 * np_), io.in(2 + 2 * np_)};
            E2 shifted = e_scalar(g_inv(io.in(0)), pt);
            WIT_OUT(shifted.a);
            WIT_OUT(shifted.b);
            E2 ev = {0, 0}, prod = {1, 0};
            auto partial = [&](u32 s, u32 e) {
                for (u32 i = s; i < e; i++) {
                    E2 term = e_sub(shifted, {dom_i, 0});
                    dom_i = g_mul(dom_i, gen);
                    E2 wv = e_scalar((u64)pr[2 + i], {io.in(1 + 2 * i), io.in(2 + 2 * i)});
                    ev = e_add(e_mul(ev, term), e_mul(wv, prod));
                    prod = e_mul(prod, term);
                }
            };
            partial(0, d);
            for (u32 i = 0; i < nint; i++) {
                WIT_OUT(ev.a);
                WIT_OUT(ev.b);
                WIT_OUT(prod.a);
                WIT_OUT(prod.b);
                u32 s = 1 + (d - 1) * (i + 1), e = s + d - 1 < np_ ? s + d - 1 : np_;
                partial(s, e);
            }
            WIT_OUT(ev.a);
            WIT_OUT(ev.b);
            break;
        }
        case OP_POSEIDON_MDS: {   // 12 extension elements in, 12 out
            const u64 circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
            u64 x[24];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int i = 0; i < 24; i++) x[i] = io.in(i);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int r = 0; r < 12; r++) {
                u64 al = 0, ah = 0, bl = 0, bh = 0;     // sums of 13 products of a 64-bit value and a 6-bit constant: 2 words
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int i = 0; i < 12; i++) {
                    int j = (i + r) % 12;
                    wit_mac128(al, ah, x[2 * j], circ[i]);
                    wit_mac128(bl, bh, x[2 * j + 1], circ[i]);
                }
                if (r == 0) {
                    wit_mac128(al, ah, x[0], 8);
                    wit_mac128(bl, bh, x[1], 8);
                }
                WIT_OUT(gl_reduce128(al, ah));
                WIT_OUT(gl_reduce128(bl, bh));
            }
            break;
        }
        case OP_REDUCING:
        case OP_REDUCING_EXT: {   // params n; in: alpha, old acc, n coefficients; out: the n accumulators (last = output)
            u32 n = (u32)pr[0];
            bool ext = op == OP_REDUCING_EXT;
            if (ni != 4 + (ext ? 2 * n : n)) return io.fail(WIT_ERR_REDUCING);
            E2 alpha = {io.in(0), io.in(1)}, acc = {io.in(2), io.in(3)};
            for (u32 i = 0; i < n; i++) {
                E2 c = ext ? E2{io.in(4 + 2 * i), io.in(5 + 2 * i)} : E2{io.in(4 + i), 0};
                acc = e_add(e_mul(acc, alpha), c);
                WIT_OUT(acc.a);
                WIT_OUT(acc.b);
            }
            break;
        }
        case OP_INTERLEAVE: {   // in x (u32); out: x with its bits spread to the even positions, then 32 big-endian bits
            const u64 x = io.in(0);
            if (x >> 32) return io.fail(WIT_ERR_INTERLEAVE);
            u64 xi = 0;
            for (int j = 0; j < 32; j++) xi |= ((x >> j) & 1) << (2 * j);
            WIT_OUT(xi);
            for (int j = 0; j < 32; j++) WIT_OUT((x >> (31 - j)) & 1);
            break;
        }
        case OP_UNINTERLEAVE: {   // param to_b32; in x; out: evens, odds, 64 big-endian bits
            const u32 step = pr[0] ? 2 : 1;
            u64 ev = 0, od = 0;
            const u64 x = io.in(0);
            for (int j = 0; j < 32; j++) {
                ev |= ((x >> (2 * j + 1)) & 1) << (step * j);
                od |= ((x >> (2 * j)) & 1) << (step * j);
            }
            WIT_OUT(ev);
            WIT_OUT(od);
            for (int j = 0; j < 64; j++) WIT_OUT((x >> (63 - j)) & 1);
            break;
        }
        default: return io.fail(WIT_ERR_OPCODE);
    }
    return true;
}
#undef WIT_OUT
