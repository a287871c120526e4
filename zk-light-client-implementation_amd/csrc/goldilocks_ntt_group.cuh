// The register-level butterfly group of the Goldilocks NTT passes (goldilocks.hip), shared with tests/hostsim.
//
// A pass does its k stages in groups of g <= 4 consecutive stages; a lane holds the M = 2^g elements of one group.  The twiddle
// of the stage-(s'+u) butterfly whose lower element has group bits m is
//     w^(((m mod 2^(g-1-u)) << (logn - g + u)) + (J << (s' + u)))  =  omega_M^((m mod 2^(g-1-u)) << u) * T^(2^u),
// omega_M = w^(n / M), T = w^(J << s'), J = the index bits below the group.  The T parts commute with the later (DIF) / earlier
// (DIT) butterflies of the group, so they collect into ONE multiplication per element, by T^(bitrev_g(m)) -- on the way out of
// a DIF group, on the way into a DIT group (M - 1 general multiplications instead of g M / 2) -- and what stays inside the group
// are powers of omega_M.  In Goldilocks 2 has order 192 and the 64th root of unity of plonky2's generator is a power of two,
//     POWER_OF_TWO_GENERATOR^(2^32 / 64) = 2^39      (omega_16 = 2^156 = -2^60, omega_8 = 2^120 = -2^24, omega_4 = 2^48),
// so every inner twiddle is a shift by a compile-time amount (2^96 = -1 folds the upper half into a swap of the subtraction's
// operands): ~14 instructions instead of a 64 x 64 multiply-reduce and a table load.
#pragma once
#include "goldilocks.cuh"
#if defined(__HIP_DEVICE_COMPILE__)
#include "goldilocks_mul_asm.inc"
#endif

#define GL_LOG2_OMEGA64 39u   // discrete logarithm of the 64th root of unity to the base 2 (checked by tests/test_hostsim_goldilocks.py)

// x * 2^e mod p for a canonical or loose x, 0 <= e < 96; e is a compile-time constant after unrolling
ZKLC_HD u64 gl_mul_2exp(u64 x, u32 e) {
    if (e == 0) return x >= GL_P ? x - GL_P : x;
    if (e < 64) return gl_reduce128(x << e, x >> (64 - e));
    if (e == 64) {                                   // x0 2^64 + x1 2^96 = x0 (2^32 - 1) - x1
        u64 a0 = x & GL_EPS, a1 = x >> 32;
        return gl_sub((a0 << 32) - a0, a1);
    }
    // e = 64 + f, 0 < f < 32: t = x << f = a0 + a1 2^32 + th 2^64 (th < 2^f);  t 2^64 = a0 (2^32 - 1) - a1 - th 2^32
    const u32 f = e - 64;
    u64 t = x << f, th = x >> (64 - f);
    u64 a0 = t & GL_EPS, a1 = t >> 32;
    return gl_sub((a0 << 32) - a0, a1 + (th << 32));  // both operands canonical: (2^32-1)^2 < p, a1 + th 2^32 < 2^63 + 2^32
}

// x[i] = x[i] * t[i] (canonical results), i < N.  On the device in batches of four / three / two independent multiplications per
// asm statement (tools/gen_gl_asm.py -> goldilocks_mul_asm.inc: 19 instructions per multiplication, carries in SGPR pairs, the
// batch interleaved so that no flag is read within two slots of its write); the compiler's gl_mul is 28.
template <int N>
ZKLC_HD void gl_mul_batch(u64 *x, const u64 *t) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKLC_GL_NO_MUL_ASM)
    int i = 0;
#define GL_LO(v) ((u32)(v))
#define GL_HI(v) ((u32)((v) >> 32))
#define GL_PACK(l, h) ((u64)(l) | ((u64)(h) << 32))
#pragma unroll
    for (; i + 4 <= N && (N - i) != 5; i += 4) {       // 5 = 3 + 2, never 4 + 1
        u32 r0l, r0h, r1l, r1h, r2l, r2h, r3l, r3h;
        gl_mul4_asm(r0l, r0h, r1l, r1h, r2l, r2h, r3l, r3h, GL_LO(x[i]), GL_HI(x[i]), GL_LO(x[i + 1]), GL_HI(x[i + 1]), GL_LO(x[i + 2]),
                    GL_HI(x[i + 2]), GL_LO(x[i + 3]), GL_HI(x[i + 3]), GL_LO(t[i]), GL_HI(t[i]), GL_LO(t[i + 1]), GL_HI(t[i + 1]),
                    GL_LO(t[i + 2]), GL_HI(t[i + 2]), GL_LO(t[i + 3]), GL_HI(t[i + 3]));
        x[i] = GL_PACK(r0l, r0h);
        x[i + 1] = GL_PACK(r1l, r1h);
        x[i + 2] = GL_PACK(r2l, r2h);
        x[i + 3] = GL_PACK(r3l, r3h);
    }
    if (N - i == 3 || N - i == 5) {
        u32 r0l, r0h, r1l, r1h, r2l, r2h;
        gl_mul3_asm(r0l, r0h, r1l, r1h, r2l, r2h, GL_LO(x[i]), GL_HI(x[i]), GL_LO(x[i + 1]), GL_HI(x[i + 1]), GL_LO(x[i + 2]), GL_HI(x[i + 2]),
                    GL_LO(t[i]), GL_HI(t[i]), GL_LO(t[i + 1]), GL_HI(t[i + 1]), GL_LO(t[i + 2]), GL_HI(t[i + 2]));
        x[i] = GL_PACK(r0l, r0h);
        x[i + 1] = GL_PACK(r1l, r1h);
        x[i + 2] = GL_PACK(r2l, r2h);
        i += 3;
    }
    if (N - i == 2) {
        u32 r0l, r0h, r1l, r1h;
        gl_mul2_asm(r0l, r0h, r1l, r1h, GL_LO(x[i]), GL_HI(x[i]), GL_LO(x[i + 1]), GL_HI(x[i + 1]), GL_LO(t[i]), GL_HI(t[i]), GL_LO(t[i + 1]),
                    GL_HI(t[i + 1]));
        x[i] = GL_PACK(r0l, r0h);
        x[i + 1] = GL_PACK(r1l, r1h);
        i += 2;
    }
    if (N - i == 1) x[i] = gl_mul(x[i], t[i]);
#undef GL_LO
#undef GL_HI
#undef GL_PACK
#else
    for (int i = 0; i < N; i++) x[i] = gl_mul(x[i], t[i]);
#endif
}

// exponent of omega_M^t as a power of two, in [0, 192)
ZKLC_HD constexpr u32 gl_omega_log2(u32 log_m, u32 t, bool inverse) {
    u32 e = (GL_LOG2_OMEGA64 * (64u >> log_m) * t) % 192u;
    return inverse ? (192u - e) % 192u : e;
}
ZKLC_HD constexpr u32 gl_bitrev_small(u32 m, int g) {
    u32 r = 0;
    for (int i = 0; i < g; i++) r |= ((m >> i) & 1u) << (g - 1 - i);
    return r;
}

// the g stages of one group on the M = 2^g values x[] of a lane.  t[m - 1] = T^(bitrev_g(m)) = w^((bitrev_g(m) * J) << s'): the
// lane's M - 1 entries of the group's table block, fetched by the caller BEFORE it reads x (all loads in flight at once; left to
// the compiler they were issued one by one in front of their multiplication, each behind its own s_waitcnt vmcnt(0)).
//
// ZP > 0 (DIF only, the FIRST group of an LDE's first pass, round 6): the top ZP index bits of the input are known to be zero (the
// coefficients of a degree-2^log_in polynomial padded to 2^(log_in + ZP)), i.e. only x[m], m < 2^(G - ZP), are non-zero on entry.
// A butterfly of stage u < ZP then has b = 0: x[m] stays, x[m | bit] = +-a * 2^e -- one shift instead of add + sub + shift -- and
// the butterflies whose two inputs are both zero are not evaluated at all (G = 4, ZP = 3: 2 + 4 + 8 shifts and the 8 butterflies
// of the last stage instead of 32 butterflies).  x[m] for m >= 2^(G - ZP) need not be initialised.
template <int G, bool DIT, bool INV, int ZP = 0>
ZKLC_HD void gl_ntt_group_regs(u64 *x, const u64 *t) {
    constexpr int M = 1 << G;
    static_assert(ZP >= 0 && ZP <= G && (ZP == 0 || !DIT), "zero-padded form: DIF groups only");
    if (DIT) gl_mul_batch<M - 1>(x + 1, t);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int uu = 0; uu < G; uu++) {
        const int u = DIT ? (G - 1 - uu) : uu;   // stage within the group; DIT runs the stages backwards
        const int bit = G - 1 - u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int m = 0; m < M; m++) {
            if (m & (1 << bit)) continue;
            const u32 lp = (u32)(m & ((1 << bit) - 1));
            const u32 e = gl_omega_log2(G, lp << u, INV);   // omega_M^(lp << u) = 2^e, e < 192
            const bool neg = e >= 96;                          // 2^96 = -1
            const u32 sh = neg ? e - 96 : e;
            if (u < ZP) {
                // the inputs that can be non-zero at this stage: bits (G - ZP) .. bit of m are all zero
                const int zmask = ((1 << (bit + 1)) - 1) & ~((1 << (G - ZP)) - 1);
                if (m & zmask) continue;
                const u64 a0 = x[m];
                x[m | (1 << bit)] = gl_mul_2exp(neg ? gl_sub(0, a0) : a0, sh);
                continue;
            }
            u64 a = x[m], b = x[m | (1 << bit)];
            if (DIT) {
                b = gl_mul_2exp(b, sh);
                x[m] = neg ? gl_sub(a, b) : gl_add(a, b);
                x[m | (1 << bit)] = neg ? gl_add(a, b) : gl_sub(a, b);
            } else {
                x[m] = gl_add(a, b);
                x[m | (1 << bit)] = gl_mul_2exp(neg ? gl_sub(b, a) : gl_sub(a, b), sh);
            }
        }
    }
    if (!DIT) gl_mul_batch<M - 1>(x + 1, t);
}

// the same group written the plain way: every butterfly with its full twiddle w^(...) (the definition above); the check of
// gl_ntt_group_regs in tests/hostsim
template <int G, bool DIT>
ZKLC_HD void gl_ntt_group_plain(u64 *x, u64 w, u32 logn, u32 s_first, u64 J) {
    constexpr int M = 1 << G;
    for (int uu = 0; uu < G; uu++) {
        const int u = DIT ? (G - 1 - uu) : uu;
        const int bit = G - 1 - u;
        for (int m = 0; m < M; m++) {
            if (m & (1 << bit)) continue;
            u64 lp = (u64)(m & ((1 << bit) - 1));
            u64 tw = gl_pow(w, (lp << (logn - G + u)) + (J << (s_first + u)));
            u64 a = x[m], b = x[m | (1 << bit)];
            if (DIT) {
                b = gl_mul(b, tw);
                x[m] = gl_add(a, b);
                x[m | (1 << bit)] = gl_sub(a, b);
            } else {
                x[m] = gl_add(a, b);
                x[m | (1 << bit)] = gl_mul(gl_sub(a, b), tw);
            }
        }
    }
}
