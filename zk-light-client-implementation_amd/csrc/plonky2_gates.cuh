// plonky2 gate-constraint evaluators over the base field (one LDE point per lane) -- the `eval_unfiltered_base`
// side of every gate type on the signature-aggregation path.  Row a9 of SURVEY section 8.
//
// Standard gates (un-vendored plonky2-near@2244a9d `plonky2/src/gates/*`), restated in the reference at
//   gnark-plonky2-verifier/plonk/gates/{arithmetic,arithmetic_extension,multiplication_extension,base_sum,constant,
//   public_input,poseidon,poseidon_mds,random_access,reducing,reducing_extension,exponentiation,coset_interpolation}_gate.go
// u32 gates (in-tree): crypto/plonky2_u32/src/gates/{arithmetic_u32,add_many_u32,subtraction_u32,range_check_u32,
//   comparison}.rs (`eval_unfiltered`).
// Filters / selector groups: plonk/gates/evaluate_gates.go:34-105.
//
// Every evaluator pushes its constraints, in the reference's order, into a consumer that keeps
// sum_i alpha_c^(offset + i) * constraint_i for each challenge c (plonk/plonk.go:186-205 reduces the full list with
// powers of alpha; the sum over gates commutes with that reduction, so no per-constraint storage is needed).
// Wires and constants are read straight from the bit-reversed LDE matrices (poly-major: column j at j * stride + p),
// so lanes of a wave read 64 consecutive u64 of one column.
#pragma once
#include "gl_ext.cuh"
#include "poseidon_gl.cuh"
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKLC_GL_NO_MUL_ASM)
#include "goldilocks_mul_asm.inc"
#endif

#define P2_MAX_CH 2
#define P2_UNUSED_SELECTOR 0xFFFFFFFFULL

enum p2_gate_type {
    P2_NOOP = 0, P2_CONSTANT, P2_PUBLIC_INPUT, P2_ARITHMETIC, P2_ARITHMETIC_EXT, P2_MUL_EXT, P2_BASE_SUM, P2_POSEIDON,
    P2_POSEIDON_MDS, P2_RANDOM_ACCESS, P2_REDUCING, P2_REDUCING_EXT, P2_EXPONENTIATION, P2_COSET_INTERPOLATION,
    P2_U32_ARITHMETIC, P2_U32_ADD_MANY, P2_U32_SUBTRACTION, P2_U32_RANGE_CHECK, P2_COMPARISON,
    P2_U32_INTERLEAVE, P2_UNINTERLEAVE_TO_U32, P2_UNINTERLEAVE_TO_B32, P2_NUM_GATE_TYPES,
    P2_POSEIDON_LAZY = 100,      // 100 + MODE, not gates of the ABI: the A/B evaluators of P2_POSEIDON (ZKLC_P2_POSEIDON_GATE=lazy | lazy1)
    P2_POSEIDON_LOOSE = 110      // ZKLC_P2_POSEIDON_GATE=loose
};

// mirrors zklc_plonky2_gate of include/zklc.h
struct p2_gate {
    u32 type;
    u32 p[4];
    u32 selector_index, group_start, group_end;
    u32 extra_off;  // offset (u64 words) into the circuit's gate_extra table
};

// sum_i alpha_c^(k0 + i) * constraint_i with the powers of alpha read from a table (wave-uniform index -> scalar loads)
// and the products accumulated unreduced in 160 bits: 2 wide multiplications per constraint instead of 4 modular ones.
struct p2_consumer {
    gl_ktab *apow[P2_MAX_CH];     // apow[c][6 k ..] = the six 22-bit limbs of alpha_c^k (gl_limbs22); scalar loads
    gl_acc3 acc[P2_MAX_CH];
    u32 k, n;
    int nch;
    ZKLC_M void reset(u32 k0) {
        k = k0;
        n = 0;
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++) acc[c].c0 = acc[c].c1 = acc[c].c2 = 0;
    }
    ZKLC_M void emit(u64 v) {
        if (++n == 480) {              // a column takes 2^10 products of 54 bits; no gate of the reference comes close
            n = 0;
#pragma unroll
            for (int c = 0; c < P2_MAX_CH; c++) gl_acc3_normalize(acc[c]);
        }
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)
            if (c < nch) gl_acc3_mul(acc[c], v, apow[c] + 6 * (size_t)k);
        k++;
    }
    ZKLC_M u64 result(int c) const { return gl_acc3_reduce(acc[c]); }
    ZKLC_M void emit2(gl2 c) {
        emit(c.a);
        emit(c.b);
    }
};

struct p2_vars {
    const u64 *wires;   // LDE of the wire polynomials, column j at wires[j * stride + p]
    const u64 *consts;  // LDE of the constants (selectors first); gate-local constant j = column nsel + j
    size_t stride, p;
    u32 nsel;
    u64 pih[4];
    ZKLC_M u64 w(u32 j) const { return wires[(size_t)j * stride + p]; }
    ZKLC_M u64 c(u32 j) const { return consts[(size_t)(nsel + j) * stride + p]; }
    ZKLC_M u64 sel(u32 j) const { return consts[(size_t)j * stride + p]; }
    ZKLC_M gl2 wa(u32 j) const { return gl2_make(w(j), w(j + 1)); }
};

#if defined(__HIPCC__)
// The same accessors over a tile staged in LDS by the fused quotient kernel (plonky2_prover.hip): column j of the tile holds
// the values of wire (or constant) j at the 64 LDE points of the workgroup, so lane l reads tile[j * 64 + l] -- one
// conflict-free ds_read_b64 per access instead of a global load per gate launch.
typedef __attribute__((address_space(3))) const u64 p2_lds_u64;
struct p2_vars_lds {
    p2_lds_u64 *wires;   // [num_wires][64]
    p2_lds_u64 *consts;  // [num_constants][64]
    u32 lane, nsel;
    u64 pih[4];
    ZKLC_M u64 w(u32 j) const { return wires[j * 64 + lane]; }
    ZKLC_M u64 c(u32 j) const { return consts[(nsel + j) * 64 + lane]; }
    ZKLC_M u64 sel(u32 j) const { return consts[j * 64 + lane]; }
    ZKLC_M gl2 wa(u32 j) const { return gl2_make(w(j), w(j + 1)); }
};
#endif

// prod_{k < base} (x - k); the result is LOOSE (any u64 congruent to it: what p2_consumer::emit multiplies by alpha^k).
// The two-bit limbs of the u32 gates dominate the constraint count of the Ed25519 circuit: x (x-1)(x-2)(x-3) = y (y + 2) with
// y = x (x - 3) takes two multiplications instead of three.
ZKLC_D u64 p2_range_product(u64 x, u32 base) {
    if (base == 4) {
        u64 y = gl_mul_loose(x, gl_sub(x, 3));
        return gl_mul_loose(y, gl_add_lc(y, 2));
    }
    u64 acc = x;
    for (u32 k = 1; k < base; k++) acc = gl_mul_loose(acc, gl_sub(x, k));
    return acc;
}
// The range checks of N two-bit limbs, r[q] = p2_range_product(x[q], 4) (loose), x canonical.  On the device in batches of four /
// three / two limbs per asm statement (tools/gen_gl_asm.py: 44 instructions per limb with the carries in SGPR pairs and the limbs
// of a batch interleaved; the compiled form is 58): the two-bit limbs are ~70 % of the constraints of the Ed25519 circuit.
template <int N>
ZKLC_D void p2_range_products4(const u64 *x, u64 *r) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKLC_GL_NO_MUL_ASM)
#define P2_LO(v) ((u32)(v))
#define P2_HI(v) ((u32)((v) >> 32))
#define P2_PACK(l, h) ((u64)(l) | ((u64)(h) << 32))
    int i = 0;
#pragma unroll
    for (; i + 4 <= N && (N - i) != 5 && (N - i) != 6; i += 4) {      // 5 = 3 + 2, 6 = 3 + 3, never 4 + 1
        u32 r0l, r0h, r1l, r1h, r2l, r2h, r3l, r3h;
        p2_range4_4_asm(r0l, r0h, r1l, r1h, r2l, r2h, r3l, r3h, P2_LO(x[i]), P2_HI(x[i]), P2_LO(x[i + 1]), P2_HI(x[i + 1]), P2_LO(x[i + 2]),
                        P2_HI(x[i + 2]), P2_LO(x[i + 3]), P2_HI(x[i + 3]));
        r[i] = P2_PACK(r0l, r0h);
        r[i + 1] = P2_PACK(r1l, r1h);
        r[i + 2] = P2_PACK(r2l, r2h);
        r[i + 3] = P2_PACK(r3l, r3h);
    }
#pragma unroll
    for (; N - i >= 3; i += 3) {
        u32 r0l, r0h, r1l, r1h, r2l, r2h;
        p2_range4_3_asm(r0l, r0h, r1l, r1h, r2l, r2h, P2_LO(x[i]), P2_HI(x[i]), P2_LO(x[i + 1]), P2_HI(x[i + 1]), P2_LO(x[i + 2]), P2_HI(x[i + 2]));
        r[i] = P2_PACK(r0l, r0h);
        r[i + 1] = P2_PACK(r1l, r1h);
        r[i + 2] = P2_PACK(r2l, r2h);
    }
    if (N - i == 2) {
        u32 r0l, r0h, r1l, r1h;
        p2_range4_2_asm(r0l, r0h, r1l, r1h, P2_LO(x[i]), P2_HI(x[i]), P2_LO(x[i + 1]), P2_HI(x[i + 1]));
        r[i] = P2_PACK(r0l, r0h);
        r[i + 1] = P2_PACK(r1l, r1h);
        i += 2;
    }
    if (N - i == 1) r[i] = p2_range_product(x[i], 4);
#undef P2_LO
#undef P2_HI
#undef P2_PACK
#else
    for (int q = 0; q < N; q++) r[q] = p2_range_product(x[q], 4);
#endif
}
// 4 * a for a loose a: (a << 2) + (a >> 62) * (2^32 - 1), one wrap possible
ZKLC_D u64 p2_mul4_loose(u64 a) {
    u64 hi = a >> 62, t = (hi << 32) - hi, r = (a << 2) + t;
    return r + ((r < t) ? GL_EPS : 0);
}
// Horner step of sum_j limb_j 4^j: loose accumulator, canonical limb
ZKLC_D u64 p2_horner4(u64 acc, u64 limb) { return gl_add_lc(p2_mul4_loose(acc), limb); }
// N consecutive wires into registers with all loads in flight at once.  The evaluators below used to read a wire, use it, read the
// next: a loop the compiler keeps rolled, one global load per iteration with its full latency exposed -- the per-gate quotient
// kernels were bound by that latency, not by HBM bandwidth or the VALU (profiles/r02_prove_ed25519_kernel_stats_v3.csv).
template <int N, class V>
ZKLC_D void p2_load(const V &v, u32 first, u64 *dst) {
#pragma unroll
    for (int q = 0; q < N; q++) dst[q] = v.w(first + q);
}
// sum of n consecutive wires (canonical), four loads in flight
template <class V>
ZKLC_D u64 p2_sum_wires(const V &v, u32 first, u32 n) {
    u64 sum = 0;
    u32 j = 0;
    for (; j + 4 <= n; j += 4) {
        u64 t[4];
        p2_load<4>(v, first + j, t);
        sum = gl_add(gl_add(sum, t[0]), gl_add(t[1], gl_add(t[2], t[3])));
    }
    for (; j < n; j++) sum = gl_add(sum, v.w(first + j));
    return sum;
}

template <class V>
ZKLC_D void p2_eval_constant(const V &v, u32 n, p2_consumer &out) {
    for (u32 i = 0; i < n; i++) out.emit(gl_sub(v.c(i), v.w(i)));
}

template <class V>
ZKLC_D void p2_eval_public_input(const V &v, p2_consumer &out) {
    for (u32 i = 0; i < 4; i++) out.emit(gl_sub(v.w(i), v.pih[i]));
}

template <class V>
ZKLC_D void p2_eval_arithmetic(const V &v, u32 num_ops, p2_consumer &out) {
    u64 c0 = v.c(0), c1 = v.c(1);
    u32 i = 0;
    for (; i + 4 <= num_ops; i += 4) {
        u64 w[16];
        p2_load<16>(v, 4 * i, w);
#pragma unroll
        for (int q = 0; q < 4; q++)
            out.emit(gl_sub(w[4 * q + 3], gl_add(gl_mul(gl_mul(w[4 * q], w[4 * q + 1]), c0), gl_mul(w[4 * q + 2], c1))));
    }
    for (; i < num_ops; i++) {
        u64 m0 = v.w(4 * i), m1 = v.w(4 * i + 1), a = v.w(4 * i + 2), o = v.w(4 * i + 3);
        out.emit(gl_sub(o, gl_add(gl_mul(gl_mul(m0, m1), c0), gl_mul(a, c1))));
    }
}

template <class V>
ZKLC_D void p2_eval_arithmetic_ext(const V &v, u32 num_ops, p2_consumer &out) {
    u64 c0 = v.c(0), c1 = v.c(1);
    for (u32 i = 0; i < num_ops; i++) {
        gl2 m0 = v.wa(8 * i), m1 = v.wa(8 * i + 2), a = v.wa(8 * i + 4), o = v.wa(8 * i + 6);
        gl2 comp = gl2_add(gl2_scale(a, c1), gl2_scale(gl2_mul(m0, m1), c0));
        out.emit2(gl2_sub(o, comp));
    }
}

template <class V>
ZKLC_D void p2_eval_mul_ext(const V &v, u32 num_ops, p2_consumer &out) {
    u64 c0 = v.c(0);
    for (u32 i = 0; i < num_ops; i++) {
        gl2 m0 = v.wa(6 * i), m1 = v.wa(6 * i + 2), o = v.wa(6 * i + 4);
        out.emit2(gl2_sub(o, gl2_scale(gl2_mul(m0, m1), c0)));
    }
}

template <class V>
ZKLC_D void p2_eval_base_sum(const V &v, u32 num_limbs, u32 base, p2_consumer &out) {
    u64 acc = 0;
    u32 i = num_limbs;
    for (; i >= 8; i -= 8) {     // Horner from the top limb, eight loads in flight
        u64 l[8];
        p2_load<8>(v, 1 + i - 8, l);
#pragma unroll
        for (int q = 7; q >= 0; q--) acc = base == 2 ? gl_add(gl_add(acc, acc), l[q]) : base == 4 ? p2_horner4(acc, l[q]) : gl_add(gl_mul(acc, base), l[q]);
    }
    while (i-- > 0) acc = base == 2 ? gl_add(gl_add(acc, acc), v.w(1 + i)) : base == 4 ? p2_horner4(acc, v.w(1 + i)) : gl_add(gl_mul(acc, base), v.w(1 + i));
    if (base == 4) acc = gl_canonical(acc);
    out.emit(gl_sub(acc, v.w(0)));
    i = 0;
    for (; i + 8 <= num_limbs; i += 8) {
        u64 l[8];
        p2_load<8>(v, 1 + i, l);
        if (base == 4) {
            u64 rp[8];
            p2_range_products4<8>(l, rp);
#pragma unroll
            for (int q = 0; q < 8; q++) out.emit(rp[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++) out.emit(p2_range_product(l[q], base));
        }
    }
    for (; i < num_limbs; i++) out.emit(p2_range_product(v.w(1 + i), base));
}

// the unrolled rounds below read ~130 dwords of table each: left alone, the scheduler hoists the scalar loads of later rounds over
// earlier ones and spills SGPRs through v_writelane / v_readlane (1 700 of them in the first build)
#if defined(__HIP_DEVICE_COMPILE__)
#define P2_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define P2_SCHED_FENCE() ((void)0)
#endif
// poseidon_gate.go:84-181, the OPT-IN form (ZKLC_P2_POSEIDON_GATE=lazy).  The same 123 constraints as p2_eval_poseidon below with
// ~20 % fewer instructions: the S-box inputs of every round but the first are WIRES, so
//   * the eight full rounds are the hand-scheduled statements of the hash kernel (poseidon_gl_asm.inc; the constraint
//     "computed - wire" and the switch to the wire value sit between two statements) -- on the host the loose C++ forms;
//   * the 22 partial rounds are not a chain: z_k = wire_k^7 is known up front, and in the lazy form of the hash kernel the value a
//     round computes for the next S-box input is a linear form  25 z_q + K_q + <u, w_q> + sum_{k<q} z_k c_q[k]  of the block's inputs --
//     carry-free multiply-accumulates over the 22-bit-limb tables PGL_LAZY_* (six v_mad_u64_u32 per product, one reduction per
//     constraint instead of one per product);
//   * constraints are emitted as loose values (the consumer multiplies 32-bit halves).
// Measured (profiles/r04t_*): proof bytes equal to the C prover's, quotient phase of the 2^18 x 234 shape 12.76 -> 12.51 ms -- a
// tenth of what the instruction count promises: the statements clobber s[32:100], the compiler keeps its long-lived scalars (table
// and alpha-power pointers) in spilled lanes (1 100 v_readlane), and the unrolled partial rounds make 110 KB of code for a 64 KB
// instruction cache.  Not the default until the partial rounds are a generated statement too.
// MODE bit 0: the partial rounds as ROLLED loops over per-lane arrays in LDS (z_k and the block's inputs u_i: a few hundred
// instructions of code instead of 8 000 unrolled ones, 60 spill instructions instead of 1 500) -- measured SLOWER (2.86 ms against
// 2.73 unrolled and 2.80 of the default: latency of the per-iteration table fetches at three waves per SIMD, 45 KB of LDS per
// workgroup); bit 1 (full rounds from the loose C++ forms instead of the statements) is not instantiated: 6.1 ms.
// profiles/r04y_poseidon_gate_variants.txt
#define P2_LAZY_THREADS 256
template <class V, int MODE>
ZKLC_D void p2_eval_poseidon_lazy(const V &v, p2_consumer &out) {
    constexpr bool ROLLED = (MODE & 1) != 0, ASM = (MODE & 2) == 0;
    u64 swap = v.w(24);
    out.emit(gl_mul(swap, gl_sub(swap, 1)));
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 lhs = v.w(i), rhs = v.w(i + 4), delta = v.w(25 + i);
        out.emit(gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), delta));
        s[i] = gl_add(lhs, delta);
        s[i + 4] = gl_sub(rhs, delta);
    }
#pragma unroll
    for (int i = 8; i < 12; i++) s[i] = v.w(i);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_lc(s[i], PGL_RC[i]);
    // eight iterations "constraints of this round's S-box inputs, then the round": it = 0..3 the first half (the 4th merged with the
    // initial matrix and followed by the partial rounds), it = 4..7 the second half; every round leaves the NEXT S-box inputs
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int it = 0; it < 8; it++) {
        if (it != 0) {
            const u32 w0 = it < 4 ? 29 + 12 * (it - 1) : 87 + 12 * (it - 4);
#pragma unroll
            for (int i = 0; i < 12; i++) {
                u64 sin = v.w(w0 + i);
                out.emit(gl_sub(s[i], sin));
                s[i] = sin;
            }
        }
        if (it != 3) {
            pgl_gate_full_round<ASM>(s, it);
            continue;
        }
        pgl_gate_full_round_init<ASM>(s);
        if constexpr (ROLLED) {
#if defined(__HIP_DEVICE_COMPILE__)
            __shared__ u64 lz_sh[22 * P2_LAZY_THREADS];
            u64 *lz = lz_sh + threadIdx.x;
#define P2_LZ(k) lz[(k) * P2_LAZY_THREADS]
#else
            u64 lz[22];
#define P2_LZ(k) lz[k]
#endif
#if defined(__HIPCC__)
#pragma unroll 1
#endif
            for (int b = 0; b < 2; b++) {
#pragma unroll
                for (int i = 0; i < 11; i++) P2_LZ(11 + i) = s[1 + i];
                u64 s0 = s[0];
#if defined(__HIPCC__)
#pragma unroll 1
#endif
                for (int q = 0; q < 11; q++) {
                    u64 sin = v.w(65 + 11 * b + q);
                    out.emit(gl_sub(s0, sin));
                    u64 zq = pgl_sbox_l(sin);
                    P2_LZ(q) = zq;
                    gl_acc3 acc;
                    const u32 kq = (11 * b + q) * 3, row = (11 * b + q) * 66;
                    acc.c0 = PGL_LAZY_K[kq] + (u64)(u32)zq * 25;
                    acc.c1 = PGL_LAZY_K[kq + 1] + (u64)(u32)(zq >> 32) * (25u << 10);
                    acc.c2 = PGL_LAZY_K[kq + 2];
#pragma unroll
                    for (int i = 0; i < 11; i++) {
                        gl_acc3_mul(acc, P2_LZ(11 + i), PGL_LAZY_W + row + 6 * i);
                        if ((i & 3) == 3) P2_SCHED_FENCE();
                    }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
                    for (int k = 0; k < q; k++) gl_acc3_mul(acc, P2_LZ(k), PGL_LAZY_C + row + 6 * k);
                    s0 = gl_acc3_reduce(acc);
                }
                s[0] = s0;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
                for (int j = 0; j < 11; j++) {
                    gl_acc3 acc;
                    const u32 kv = (11 * b + j) * 3;
                    const u64 uj = P2_LZ(11 + j);
                    acc.c0 = PGL_LAZY_KV[kv] + (u64)(u32)uj;
                    acc.c1 = PGL_LAZY_KV[kv + 1] + ((u64)(u32)(uj >> 32) << 10);
                    acc.c2 = PGL_LAZY_KV[kv + 2];
#pragma unroll
                    for (int k = 0; k < 11; k++) {
                        gl_acc3_mul(acc, P2_LZ(k), PGL_LAZY_V + (11 * b + j) * 66 + 6 * k);
                        if ((k & 3) == 3) P2_SCHED_FENCE();
                    }
                    P2_LZ(11 + j) = gl_acc3_reduce(acc);
                }
#pragma unroll
                for (int i = 0; i < 11; i++) s[1 + i] = P2_LZ(11 + i);
            }
#undef P2_LZ
            continue;
        }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (int b = 0; b < 2; b++) {
            u64 z[11];
#if defined(__HIPCC__)
#pragma clang loop unroll(full)
#endif
            for (int q = 0; q < 11; q++) {
                u64 sin = v.w(65 + 11 * b + q);
                out.emit(gl_sub(s[0], sin));
                z[q] = pgl_sbox_l(sin);
                gl_acc3 acc;
                const u32 kq = (11 * b + q) * 3, row = (11 * b + q) * 66;
                acc.c0 = PGL_LAZY_K[kq] + (u64)(u32)z[q] * 25;
                acc.c1 = PGL_LAZY_K[kq + 1] + (u64)(u32)(z[q] >> 32) * (25u << 10);      // 2^32 = 2^22 2^10
                acc.c2 = PGL_LAZY_K[kq + 2];
#pragma unroll
                for (int i = 0; i < 11; i++) {
                    gl_acc3_mul(acc, s[1 + i], PGL_LAZY_W + row + 6 * i);
                    if ((i & 3) == 3) P2_SCHED_FENCE();      // table fetches in groups of four slots (24 SGPRs), like the hash kernel's
                }
#pragma unroll
                for (int k = 0; k < q; k++) {
                    gl_acc3_mul(acc, z[k], PGL_LAZY_C + row + 6 * k);
                    if ((k & 3) == 3) P2_SCHED_FENCE();
                }
                s[0] = gl_acc3_reduce(acc);
                P2_SCHED_FENCE();          // keep the table fetches of a round inside the round (SGPR budget)
            }
#pragma unroll
            for (int j = 0; j < 11; j++) {
                gl_acc3 acc;
                const u32 kv = (11 * b + j) * 3;
                acc.c0 = PGL_LAZY_KV[kv] + (u64)(u32)s[1 + j];
                acc.c1 = PGL_LAZY_KV[kv + 1] + ((u64)(u32)(s[1 + j] >> 32) << 10);
                acc.c2 = PGL_LAZY_KV[kv + 2];
#pragma unroll
                for (int k = 0; k < 11; k++) {
                    gl_acc3_mul(acc, z[k], PGL_LAZY_V + (11 * b + j) * 66 + 6 * k);
                    if ((k & 3) == 3) P2_SCHED_FENCE();
                }
                s[1 + j] = gl_acc3_reduce(acc);
                P2_SCHED_FENCE();
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 12; i++) out.emit(gl_sub(s[i], v.w(12 + i)));
}

// poseidon_gate.go:84-181, round by round like the default evaluator below but in the LOOSE arithmetic of the C++ permutation
// (poseidon_gl.cuh: no canonicalisation between operations, the dot products of the partial rounds accumulated in 160 bits and
// reduced once, constraints emitted as loose values).  Same loop structure and code size as the default.  Opt-in
// (ZKLC_P2_POSEIDON_GATE=loose) until it has been through the whole GPU suite.
template <class V>
ZKLC_D void p2_eval_poseidon_loose(const V &v, p2_consumer &out) {
    u64 swap = v.w(24);
    out.emit(gl_mul(swap, gl_sub(swap, 1)));
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 lhs = v.w(i), rhs = v.w(i + 4), delta = v.w(25 + i);
        out.emit(gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), delta));
        s[i] = gl_add(lhs, delta);
        s[i + 4] = gl_sub(rhs, delta);
    }
#pragma unroll
    for (int i = 8; i < 12; i++) s[i] = v.w(i);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_add_lc(s[i], PGL_RC[12 * r + i]);
        if (r != 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                u64 sin = v.w(29 + 12 * (r - 1) + i);
                out.emit(gl_sub(s[i], sin));
                s[i] = sin;
            }
        }
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = pgl_sbox_l(s[i]);
        pgl_mds_l(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_lc(s[i], PGL_FP_FIRST[i]);
    {
        // t[d] = sum_{r=1..11} s[r] * INIT[r-1][d-1]; one output per iteration, rotated into place (static register indices)
        u64 t[12];
#pragma unroll
        for (int d = 1; d < 12; d++) t[d] = 0;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (int d = 1; d < 12; d++) {
            gl_acc160 acc = {0, 0, 0};
#pragma unroll
            for (int r = 1; r < 12; r++) gl_acc_mul(acc, s[r], PGL_FP_INIT[(r - 1) * 11 + d - 1]);
#pragma unroll
            for (int q = 1; q < 11; q++) t[q] = t[q + 1];
            t[11] = gl_acc_reduce(acc);
        }
#pragma unroll
        for (int i = 1; i < 12; i++) s[i] = t[i];
    }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 22; r++) {
        u64 sin = v.w(65 + r);
        out.emit(gl_sub(s[0], sin));
        u64 s0 = gl_add_lc(pgl_sbox_l(sin), PGL_FP_RC[r]);  // the 22nd constant is zero (poseidon_gate.go:151-155)
        gl_acc160 acc = {0, 0, 0};
        gl_acc_mul(acc, s0, 25);
#pragma unroll
        for (int j = 1; j < 12; j++) gl_acc_mul(acc, s[j], PGL_FP_WHATS[r * 11 + j - 1]);
#pragma unroll
        for (int j = 1; j < 12; j++) {
            u64 lo, hi;                                      // s[j] + s0 * v < 2^128: one reduction of the sum
            gl_mul_wide(s0, PGL_FP_VS[r * 11 + j - 1], lo, hi);
            u64 l2 = lo + s[j];
            hi += (l2 < lo);
            s[j] = gl_reduce128_loose(l2, hi);
        }
        s[0] = gl_acc_reduce(acc);
    }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            u64 x = gl_add_lc(s[i], PGL_RC[12 * (26 + r) + i]);
            u64 sin = v.w(87 + 12 * r + i);
            out.emit(gl_sub(x, sin));
            s[i] = pgl_sbox_l(sin);
        }
        pgl_mds_l(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) out.emit(gl_sub(s[i], v.w(12 + i)));
}

// poseidon_gate.go:84-181, round by round in canonical arithmetic: the default evaluator
template <class V>
ZKLC_D void p2_eval_poseidon(const V &v, p2_consumer &out) {
    u64 swap = v.w(24);
    out.emit(gl_mul(swap, gl_sub(swap, 1)));
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 lhs = v.w(i), rhs = v.w(i + 4), delta = v.w(25 + i);
        out.emit(gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), delta));
        s[i] = gl_add(lhs, delta);
        s[i + 4] = gl_sub(rhs, delta);
    }
#pragma unroll
    for (int i = 8; i < 12; i++) s[i] = v.w(i);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], PGL_RC[12 * r + i]);
        if (r != 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                u64 sin = v.w(29 + 12 * (r - 1) + i);
                out.emit(gl_sub(s[i], sin));
                s[i] = sin;
            }
        }
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = pgl_sbox(s[i]);
        pgl_mds(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], PGL_FP_FIRST[i]);
    {
        u64 t[12];
        t[0] = s[0];
#pragma unroll
        for (int d = 1; d < 12; d++) t[d] = 0;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (int r = 1; r < 12; r++) {
            u64 sr = s[1];
#pragma unroll
            for (int d = 1; d < 12; d++) t[d] = gl_add(t[d], gl_mul(sr, PGL_FP_INIT[(r - 1) * 11 + d - 1]));
#pragma unroll
            for (int q = 1; q < 11; q++) s[q] = s[q + 1];
        }
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = t[i];
    }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 22; r++) {
        u64 sin = v.w(65 + r);
        out.emit(gl_sub(s[0], sin));
        u64 s0 = gl_add(pgl_sbox(sin), PGL_FP_RC[r]);  // the 22nd constant is zero (poseidon_gate.go:151-155)
        u64 d = gl_mul(s0, 25);
#pragma unroll
        for (int j = 1; j < 12; j++) d = gl_add(d, gl_mul(s[j], PGL_FP_WHATS[r * 11 + j - 1]));
#pragma unroll
        for (int j = 1; j < 12; j++) s[j] = gl_add(s[j], gl_mul(s0, PGL_FP_VS[r * 11 + j - 1]));
        s[0] = d;
    }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            u64 x = gl_add(s[i], PGL_RC[12 * (26 + r) + i]);
            u64 sin = v.w(87 + 12 * r + i);
            out.emit(gl_sub(x, sin));
            s[i] = pgl_sbox(sin);
        }
        pgl_mds(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) out.emit(gl_sub(s[i], v.w(12 + i)));
}

template <class V>
ZKLC_D void p2_eval_poseidon_mds(const V &v, p2_consumer &out) {
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    for (u32 r = 0; r < 12; r++) {
        gl2 acc = gl2_make(0, 0);
        for (u32 i = 0; i < 12; i++) acc = gl2_add(acc, gl2_scale(v.wa(2 * ((i + r) % 12)), C[i]));
        if (r == 0) acc = gl2_add(acc, gl2_scale(v.wa(0), 8));
        out.emit2(gl2_sub(v.wa(2 * (12 + r)), acc));
    }
}

template <class V>
ZKLC_D void p2_eval_random_access(const V &v, u32 bits, u32 copies, u32 extra, p2_consumer &out) {
    const u32 vs = 1u << bits;
    const u32 routed = (2 + vs) * copies + extra;
    for (u32 cp = 0; cp < copies; cp++) {
        u32 base = (2 + vs) * cp;
        u64 idx = v.w(base), claimed = v.w(base + 1);
        u64 rec = 0;
        for (u32 i = 0; i < bits; i++) {
            u64 b = v.w(routed + cp * bits + i);
            out.emit(gl_sub(gl_sqr(b), b));
        }
        for (u32 i = bits; i-- > 0;) rec = gl_add(gl_double(rec), v.w(routed + cp * bits + i));
        out.emit(gl_sub(rec, idx));
        // fold the list by the bits (lowest bit first): item = x + b (y - x); at most 64 entries (bits <= 6)
        u64 items[64];
        for (u32 i = 0; i < vs; i++) items[i] = v.w(base + 2 + i);
        u32 len = vs;
        for (u32 i = 0; i < bits; i++) {
            u64 b = v.w(routed + cp * bits + i);
            len >>= 1;
            for (u32 k = 0; k < len; k++) items[k] = gl_add(items[2 * k], gl_mul(b, gl_sub(items[2 * k + 1], items[2 * k])));
        }
        out.emit(gl_sub(items[0], claimed));
    }
    for (u32 i = 0; i < extra; i++) out.emit(gl_sub(v.c(i), v.w((2 + vs) * copies + i)));
}

template <class V>
ZKLC_D void p2_eval_reducing(const V &v, u32 n, bool ext, p2_consumer &out) {
    gl2 alpha = v.wa(2), acc = v.wa(4);
    const u32 start_accs = 6 + (ext ? 2 * n : n);
    for (u32 i = 0; i < n; i++) {
        gl2 nxt = (i == n - 1) ? v.wa(0) : v.wa(start_accs + 2 * i);
        gl2 coeff = ext ? v.wa(6 + 2 * i) : gl2_make(v.w(6 + i), 0);
        out.emit2(gl2_sub(gl2_add(gl2_mul(acc, alpha), coeff), nxt));
        acc = nxt;
    }
}

template <class V>
ZKLC_D void p2_eval_exponentiation(const V &v, u32 n, p2_consumer &out) {
    u64 base = v.w(0);
    u64 prev_inter = 0;
    for (u32 i = 0; i < n; i++) {
        u64 prev = i == 0 ? 1 : gl_sqr(prev_inter);
        u64 b = v.w(1 + (n - 1 - i));
        u64 mul_by = gl_sub(gl_mul(b, base), gl_sub(b, 1));
        u64 inter = v.w(2 + n + i);
        out.emit(gl_sub(gl_mul(prev, mul_by), inter));
        prev_inter = inter;
    }
    out.emit(gl_sub(v.w(1 + n), prev_inter));
}

// coset_interpolation_gate.go:152-226; extra = [barycentric weights (2^bits) | subgroup points w^i (2^bits)]
template <class V>
ZKLC_D void p2_coset_partial(const V &v, const u64 *extra, u32 np, u32 s, u32 e, gl2 point, gl2 &ev, gl2 &prod) {
    for (u32 i = s; i < e; i++) {
        gl2 term = gl2_sub(point, gl2_make(extra[np + i], 0));
        gl2 wv = gl2_scale(v.wa(1 + 2 * i), extra[i]);
        ev = gl2_add(gl2_mul(ev, term), gl2_mul(wv, prod));
        prod = gl2_mul(prod, term);
    }
}
template <class V>
ZKLC_D void p2_eval_coset_interpolation(const V &v, u32 bits, u32 degree, const u64 *extra, p2_consumer &out) {
    const u32 np = 1u << bits;
    const u32 n_inter = (np - 2) / (degree - 1);
    const u32 start_pt = 1 + 2 * np, start_val = start_pt + 2, start_inter = start_val + 2;
    u64 shift = v.w(0);
    gl2 point = v.wa(start_pt), shifted = v.wa(start_inter + 4 * n_inter);
    out.emit2(gl2_add(gl2_scale(shifted, gl_neg(shift)), point));
    gl2 ev = gl2_make(0, 0), prod = gl2_make(1, 0);
    p2_coset_partial(v, extra, np, 0, degree, shifted, ev, prod);
    for (u32 i = 0; i < n_inter; i++) {
        gl2 iev = v.wa(start_inter + 2 * i), ipr = v.wa(start_inter + 2 * (n_inter + i));
        out.emit2(gl2_sub(iev, ev));
        out.emit2(gl2_sub(ipr, prod));
        u32 s = 1 + (degree - 1) * (i + 1);
        u32 e = s + degree - 1 < np ? s + degree - 1 : np;
        ev = iev;
        prod = ipr;
        p2_coset_partial(v, extra, np, s, e, shifted, ev, prod);
    }
    out.emit2(gl2_sub(v.wa(start_val), ev));
}

// crypto/plonky2_u32/src/gates/arithmetic_u32.rs:110-165
template <class V>
ZKLC_D void p2_eval_u32_arithmetic(const V &v, u32 num_ops, p2_consumer &out) {
    for (u32 i = 0; i < num_ops; i++) {
        u64 r[6], lh[16], ll[16];
        p2_load<6>(v, 6 * i, r);
        p2_load<16>(v, 6 * num_ops + 32 * i + 16, lh);
        p2_load<16>(v, 6 * num_ops + 32 * i, ll);
        u64 m0 = r[0], m1 = r[1], add = r[2], lo = r[3], hi = r[4], inv = r[5];
        u64 computed = gl_add(gl_mul(m0, m1), add);
        u64 diff = gl_sub(0xFFFFFFFFULL, hi);
        u64 hi_not_max = gl_sub(gl_mul(inv, diff), 1);
        out.emit(gl_mul(hi_not_max, lo));
        out.emit(gl_sub(gl_add(gl_mul(hi, 1ULL << 32), lo), computed));
        u64 comb_lo = 0, comb_hi = 0;
        u64 rev[16], rp[16];
#pragma unroll
        for (int j = 0; j < 16; j++) rev[j] = lh[15 - j];
        p2_range_products4<16>(rev, rp);
#pragma unroll
        for (int j = 15; j >= 0; j--) {
            out.emit(rp[15 - j]);
            comb_hi = p2_horner4(comb_hi, lh[j]);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) rev[j] = ll[15 - j];
        p2_range_products4<16>(rev, rp);
#pragma unroll
        for (int j = 15; j >= 0; j--) {
            out.emit(rp[15 - j]);
            comb_lo = p2_horner4(comb_lo, ll[j]);
        }
        out.emit(gl_sub(gl_canonical(comb_lo), lo));
        out.emit(gl_sub(gl_canonical(comb_hi), hi));
    }
}

// add_many_u32.rs: per op (num_addends + 3) routed wires, then 16 + 2 two-bit limbs
template <class V>
ZKLC_D void p2_eval_u32_add_many(const V &v, u32 num_addends, u32 num_ops, p2_consumer &out) {
    const u32 per = num_addends + 3;
    for (u32 i = 0; i < num_ops; i++) {
        u64 l[18], rc[2];
        p2_load<18>(v, per * num_ops + 18 * i, l);
        p2_load<2>(v, per * i + num_addends + 1, rc);
        u64 sum = p2_sum_wires(v, per * i, num_addends + 1);   // the addends and the carry in
        u64 res = rc[0], carry = rc[1];
        out.emit(gl_sub(gl_add(gl_mul(carry, 1ULL << 32), res), sum));
        u64 comb_res = 0, comb_carry = 0;
        u64 rev[18], rp[18];
#pragma unroll
        for (int j = 0; j < 18; j++) rev[j] = l[17 - j];
        p2_range_products4<18>(rev, rp);
#pragma unroll
        for (int j = 17; j >= 0; j--) {
            out.emit(rp[17 - j]);
            if (j < 16)
                comb_res = p2_horner4(comb_res, l[j]);
            else
                comb_carry = p2_horner4(comb_carry, l[j]);
        }
        out.emit(gl_sub(gl_canonical(comb_res), res));
        out.emit(gl_sub(gl_canonical(comb_carry), carry));
    }
}

// ---- the lane-level pieces of the U32AddMany LDS-tile evaluator (plonky2_prover.hip: p2_quotient_addmany_tile_kernel), shared with
// tests/hostsim.  S = the wave's variant slot (static: the accumulators stay in registers).
// routed wires of one variant: the sum constraint and the -res / -carry halves of the two recombination constraints
template <int S, class V>
ZKLC_D void p2_amt_routed(gl_acc3 (&acc)[2][P2_MAX_CH], const V &pv, u32 na, u32 ops, gl_ktab *const (&apow)[P2_MAX_CH], int nch, u32 k0) {
    const u32 per = na + 3;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (u32 i = 0; i < ops; i++) {
        u64 rc[2];
        p2_load<2>(pv, per * i + na + 1, rc);
        const u64 sum = p2_sum_wires(pv, per * i, na + 1);          // the addends and the carry in
        const u64 e0 = gl_sub(gl_add(gl_mul(rc[1], 1ULL << 32), rc[0]), sum), e1 = gl_neg(rc[0]), e2 = gl_neg(rc[1]);
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)
            if (c < nch) {
                gl_ktab *t = apow[c] + 6 * (size_t)(k0 + 21 * i);
                gl_acc3_mul(acc[S][c], e0, t);
                gl_acc3_mul(acc[S][c], e1, t + 6 * 19);
                gl_acc3_mul(acc[S][c], e2, t + 6 * 20);
            }
    }
}
// the part of variant (ops, l0)'s limb region inside the phase [base, top), columns descending; S = the wave's variant slot
template <int S>
ZKLC_D void p2_amt_consume(gl_acc3 (&acc)[2][P2_MAX_CH], u64 (&comb)[2], u32 ops, u32 l0, u32 base, u32 top, const u64 *tw,
                           const u64 *trp, u32 lane, gl_ktab *const (&apow)[P2_MAX_CH], int nch, u32 k0) {
    const u32 r_end = l0 + 18 * ops;
    const u32 c_hi = top < r_end ? top : r_end, c_lo = base > l0 ? base : l0;
    if (c_hi <= c_lo) return;                          // the region misses the phase
    auto emit = [&](u32 krel, u64 val) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)
            if (c < nch) gl_acc3_mul(acc[S][c], val, apow[c] + 6 * (size_t)(k0 + krel));
    };
    // one Horner step of the limb recombination; after limb 16 (carry part complete) and limb 0 (result part complete) the sum is
    // emitted at its constraint and restarted.  The running sum lives in a LOCAL for the whole call (written back once): as
    // `comb[S] = 0` inside the conditional the restart after limb 0 was lost by the device compiler (round 4, found by dumping the
    // emitted (constraint, value) pairs of one lane: every carry sum still held the previous operation's result sum).
    u64 cb = comb[S];
    auto horner = [&](u64 w, u32 i, u32 lj) __attribute__((always_inline)) {
        cb = p2_horner4(cb, w);
        const bool carry_done = lj == 16, res_done = lj == 0;
        if (carry_done | res_done) {
            emit(21 * i + (carry_done ? 20u : 19u), gl_canonical(cb));
            cb = 0;
        }
    };
    u32 c = c_hi;                                      // exclusive
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    while (c > c_lo) {
        const u32 rel = c - 1 - l0, i = rel / 18, l = rel - 18 * i;     // the top column of this run is limb l of operation i
        u32 cnt = l + 1;                               // down to limb 0 of the operation, or to the bottom of the phase / region
        if (cnt > c - c_lo) cnt = c - c_lo;
        const u32 kb = 21 * i + 18 - l;                // constraint of limb l; limb l - q is constraint kb + q
        u32 q = 0;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (; q + 4 <= cnt; q += 4) {
            u64 w[4], rp[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 col = c - 1 - q - j - base;
                w[j] = tw[(size_t)col * 64 + lane];
                rp[j] = trp[(size_t)col * 64 + lane];
            }
#pragma unroll
            for (int ch = 0; ch < P2_MAX_CH; ch++)
                if (ch < nch) {
                    gl_ktab *t = apow[ch] + 6 * (size_t)(k0 + kb + q);
#pragma unroll
                    for (int j = 0; j < 4; j++) gl_acc3_mul(acc[S][ch], rp[j], t + 6 * j);
                }
#pragma unroll
            for (int j = 0; j < 4; j++) horner(w[j], i, l - q - j);
        }
#if defined(__HIPCC__)
#pragma unroll 1
#endif
        for (; q < cnt; q++) {
            const u32 col = c - 1 - q - base;
            const u64 w = tw[(size_t)col * 64 + lane], rp = trp[(size_t)col * 64 + lane];
            emit(kb + q, rp);
            horner(w, i, l - q);
        }
        c -= cnt;
    }
    comb[S] = cb;
}

// subtraction_u32.rs: per op (x, y, borrow_in, result, borrow_out), then 16 two-bit limbs of result
template <class V>
ZKLC_D void p2_eval_u32_subtraction(const V &v, u32 num_ops, p2_consumer &out) {
    for (u32 i = 0; i < num_ops; i++) {
        u64 r[5], l[16];
        p2_load<5>(v, 5 * i, r);
        p2_load<16>(v, 5 * num_ops + 16 * i, l);
        u64 x = r[0], y = r[1], bin = r[2], res = r[3], bout = r[4];
        u64 initial = gl_sub(gl_sub(x, y), bin);
        out.emit(gl_sub(res, gl_add(initial, gl_mul(bout, 1ULL << 32))));
        u64 comb = 0;
        u64 rev[16], rp[16];
#pragma unroll
        for (int j = 0; j < 16; j++) rev[j] = l[15 - j];
        p2_range_products4<16>(rev, rp);
#pragma unroll
        for (int j = 15; j >= 0; j--) {
            out.emit(rp[15 - j]);
            comb = p2_horner4(comb, l[j]);
        }
        out.emit(gl_sub(gl_canonical(comb), res));
        out.emit(gl_mul(bout, gl_sub(1, bout)));
    }
}

// range_check_u32.rs: n input limbs, each with 16 two-bit aux limbs
template <class V>
ZKLC_D void p2_eval_u32_range_check(const V &v, u32 n, p2_consumer &out) {
    for (u32 i = 0; i < n; i++) {
        u64 l[16];
        p2_load<16>(v, n + 16 * i, l);
        u64 sum = 0;
#pragma unroll
        for (int j = 15; j >= 0; j--) sum = p2_horner4(sum, l[j]);
        out.emit(gl_sub(gl_canonical(sum), v.w(i)));
        u64 rp[16];
        p2_range_products4<16>(l, rp);
#pragma unroll
        for (int j = 0; j < 16; j++) out.emit(rp[j]);
    }
}

// comparison.rs:106-190
template <class V>
ZKLC_D void p2_eval_comparison(const V &v, u32 num_bits, u32 num_chunks, p2_consumer &out) {
    const u32 chunk_bits = (num_bits + num_chunks - 1) / num_chunks;
    const u32 chunk_size = 1u << chunk_bits;
    u64 c1 = 0, c2 = 0;
    {
        u32 i = num_chunks;
        for (; i >= 8; i -= 8) {     // both Horner sums from the top chunk, sixteen loads in flight
            u64 a[8], b[8];
            p2_load<8>(v, 4 + i - 8, a);
            p2_load<8>(v, 4 + num_chunks + i - 8, b);
#pragma unroll
            for (int q = 7; q >= 0; q--) {
                c1 = chunk_size == 4 ? p2_horner4(c1, a[q]) : gl_add(gl_mul(c1, chunk_size), a[q]);
                c2 = chunk_size == 4 ? p2_horner4(c2, b[q]) : gl_add(gl_mul(c2, chunk_size), b[q]);
            }
        }
        while (i-- > 0) {
            c1 = chunk_size == 4 ? p2_horner4(c1, v.w(4 + i)) : gl_add(gl_mul(c1, chunk_size), v.w(4 + i));
            c2 = chunk_size == 4 ? p2_horner4(c2, v.w(4 + num_chunks + i)) : gl_add(gl_mul(c2, chunk_size), v.w(4 + num_chunks + i));
        }
        if (chunk_size == 4) {
            c1 = gl_canonical(c1);
            c2 = gl_canonical(c2);
        }
    }
    out.emit(gl_sub(c1, v.w(0)));
    out.emit(gl_sub(c2, v.w(1)));
    u64 msd = 0;
    auto chunk = [&](u64 a, u64 b, u64 dummy, u64 eq, u64 inter) {
        out.emit(p2_range_product(a, chunk_size));
        out.emit(p2_range_product(b, chunk_size));
        u64 diff = gl_sub(b, a);
        out.emit(gl_sub(gl_mul(diff, dummy), gl_sub(1, eq)));
        out.emit(gl_mul(eq, diff));
        out.emit(gl_sub(inter, gl_mul(eq, msd)));
        msd = gl_add(inter, gl_mul(gl_sub(1, eq), diff));
    };
    u32 i = 0;
    for (; i + 4 <= num_chunks; i += 4) {     // twenty loads in flight
        u64 a[4], b[4], d[4], e[4], in[4];
        p2_load<4>(v, 4 + i, a);
        p2_load<4>(v, 4 + num_chunks + i, b);
        p2_load<4>(v, 4 + 2 * num_chunks + i, d);
        p2_load<4>(v, 4 + 3 * num_chunks + i, e);
        p2_load<4>(v, 4 + 4 * num_chunks + i, in);
#pragma unroll
        for (int q = 0; q < 4; q++) chunk(a[q], b[q], d[q], e[q], in[q]);
    }
    for (; i < num_chunks; i++)
        chunk(v.w(4 + i), v.w(4 + num_chunks + i), v.w(4 + 2 * num_chunks + i), v.w(4 + 3 * num_chunks + i), v.w(4 + 4 * num_chunks + i));
    u64 msd_w = v.w(3);
    out.emit(gl_sub(msd_w, msd));
    u64 comb = 0;
    for (u32 i = 0; i <= chunk_bits; i++) {
        u64 bit = v.w(4 + 5 * num_chunks + i);
        out.emit(gl_mul(bit, gl_sub(1, bit)));
    }
    for (u32 i = chunk_bits + 1; i-- > 0;) comb = gl_add(gl_double(comb), v.w(4 + 5 * num_chunks + i));
    out.emit(gl_sub(gl_add(chunk_size, msd_w), comb));
    out.emit(gl_sub(v.w(2), v.w(4 + 5 * num_chunks + chunk_bits)));
}

// interleave_u32.rs:103-139: per op (x, x_interleaved) routed, then 32 big-endian bits; x = sum bits 2^(31-j),
// x_interleaved = sum bits 4^(31-j) (the bits of x spread over the even positions of a 64-bit word)
template <class V>
ZKLC_D void p2_eval_u32_interleave(const V &v, u32 num_ops, p2_consumer &out) {
    for (u32 i = 0; i < num_ops; i++) {
        u64 x = 0, xi = 0;
        for (u32 j = 0; j < 32; j++) {
            u64 b = v.w(2 * num_ops + 32 * i + j);
            x = gl_add(gl_add(x, x), b);
            xi = p2_horner4(xi, b);
        }
        out.emit(gl_sub(x, v.w(2 * i)));
        out.emit(gl_sub(gl_canonical(xi), v.w(2 * i + 1)));
        for (u32 j = 0; j < 32; j++) out.emit(p2_range_product(v.w(2 * num_ops + 32 * i + j), 2));
    }
}
// uninterleave_to_u32.rs:112-159 / uninterleave_to_b32.rs: per op (x_interleaved, evens, odds) routed, then 64 big-endian bits;
// evens / odds collect bits 2j / 2j+1 with weights 2^(31-j) (to_u32) or 4^(31-j) (to_b32: the halves stay interleaved)
template <class V>
ZKLC_D void p2_eval_uninterleave(const V &v, u32 num_ops, bool to_b32, p2_consumer &out) {
    for (u32 i = 0; i < num_ops; i++) {
        u64 x = 0, ev = 0, od = 0;
        for (u32 j = 0; j < 32; j++) {
            u64 be = v.w(3 * num_ops + 64 * i + 2 * j), bo = v.w(3 * num_ops + 64 * i + 2 * j + 1);
            x = gl_add(gl_add(gl_add(x, x), be), gl_add(gl_add(gl_add(x, x), be), bo));   // x = 4x + 2 be + bo
            if (to_b32) {
                ev = p2_horner4(ev, be);
                od = p2_horner4(od, bo);
            } else {
                ev = gl_add(gl_add(ev, ev), be);
                od = gl_add(gl_add(od, od), bo);
            }
        }
        out.emit(gl_sub(x, v.w(3 * i)));
        out.emit(gl_sub(gl_canonical(ev), v.w(3 * i + 1)));
        out.emit(gl_sub(gl_canonical(od), v.w(3 * i + 2)));
        for (u32 j = 0; j < 64; j++) out.emit(p2_range_product(v.w(3 * num_ops + 64 * i + j), 2));
    }
}

template <class V>
ZKLC_D void p2_eval_gate(const p2_gate &g, const V &v, const u64 *extra, p2_consumer &out) {
    switch (g.type) {
        case P2_NOOP: break;
        case P2_CONSTANT: p2_eval_constant(v, g.p[0], out); break;
        case P2_PUBLIC_INPUT: p2_eval_public_input(v, out); break;
        case P2_ARITHMETIC: p2_eval_arithmetic(v, g.p[0], out); break;
        case P2_ARITHMETIC_EXT: p2_eval_arithmetic_ext(v, g.p[0], out); break;
        case P2_MUL_EXT: p2_eval_mul_ext(v, g.p[0], out); break;
        case P2_BASE_SUM: p2_eval_base_sum(v, g.p[0], g.p[1], out); break;
        case P2_POSEIDON: p2_eval_poseidon(v, out); break;
        case P2_POSEIDON_MDS: p2_eval_poseidon_mds(v, out); break;
        case P2_RANDOM_ACCESS: p2_eval_random_access(v, g.p[0], g.p[1], g.p[2], out); break;
        case P2_REDUCING: p2_eval_reducing(v, g.p[0], false, out); break;
        case P2_REDUCING_EXT: p2_eval_reducing(v, g.p[0], true, out); break;
        case P2_EXPONENTIATION: p2_eval_exponentiation(v, g.p[0], out); break;
        case P2_COSET_INTERPOLATION: p2_eval_coset_interpolation(v, g.p[0], g.p[1], extra + g.extra_off, out); break;
        case P2_U32_ARITHMETIC: p2_eval_u32_arithmetic(v, g.p[0], out); break;
        case P2_U32_ADD_MANY: p2_eval_u32_add_many(v, g.p[0], g.p[1], out); break;
        case P2_U32_SUBTRACTION: p2_eval_u32_subtraction(v, g.p[0], out); break;
        case P2_U32_RANGE_CHECK: p2_eval_u32_range_check(v, g.p[0], out); break;
        case P2_COMPARISON: p2_eval_comparison(v, g.p[0], g.p[1], out); break;
        case P2_U32_INTERLEAVE: p2_eval_u32_interleave(v, g.p[0], out); break;
        case P2_UNINTERLEAVE_TO_U32: p2_eval_uninterleave(v, g.p[0], false, out); break;
        case P2_UNINTERLEAVE_TO_B32: p2_eval_uninterleave(v, g.p[0], true, out); break;
        default: break;
    }
}

// evaluate_gates.go:34-57: prod_{i in group, i != row} (i - s) [* (UNUSED - s) when there are several selectors]
ZKLC_D u64 p2_filter(u32 row, u32 start, u32 end, u64 s, bool many) {
    u64 prod = 1;
    for (u32 i = start; i < end; i++)
        if (i != row) prod = gl_mul(prod, gl_sub(i, s));
    if (many) prod = gl_mul(prod, gl_sub(P2_UNUSED_SELECTOR, s));
    return prod;
}
