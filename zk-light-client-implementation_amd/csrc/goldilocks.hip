// Goldilocks NTT / LDE, Poseidon permutation, leaf hashing and Merkle trees
// (gfx950) + their C ABI.  These are the kernels behind the plonky2 commit phase
// (`PolynomialBatch::from_values/from_coeffs`, `MerkleTree::new` in the
// un-vendored plonky2-near fork; reference call sites
// near_bft_finality/src/prove_crypto/ed25519.rs:60,100, recursion.rs:95).
//
// Data layout in HBM: a batch of polynomials is POLY-MAJOR (poly p, index i at
// p * stride + i).  That is the natural layout for the NTT (each transform is
// contiguous) AND for leaf hashing with one lane per leaf (for every column p the
// 64 lanes of a wave read 64 consecutive u64 = 512 contiguous bytes), so the
// "transpose to row-major leaves" pass of the CPU prover disappears.
#include <stdlib.h>
#include "poseidon_gl.cuh"
#include "goldilocks_ntt_group.cuh"
#include <mutex>
#include "zklc_internal.h"

#define NTT_THREADS 256
#define NTT_TILE_LOG_MAX 12
#define NTT_TILE_MAX (1 << NTT_TILE_LOG_MAX)
// the radix-8 pass kernel takes its tile as dynamic LDS: 2^12 elements (36 KB with padding, 256 threads) or 2^13 (72 KB, 512 threads)
#define NTT_TILE_LOG_BIG 13
#define NTT_THREADS_MAX 512

// ---------------------------------------------------------------- tables
__global__ void gl_pow_table_kernel(u64 *out, u64 base, u64 n, u64 exp_stride) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl_pow(base, i * exp_stride);
}

// staged twiddles: for every stage s the values w^(j << s), j < n / 2^(s+1), contiguous at offset n - (n >> s)
// (n entries in total).  Lanes with consecutive j read consecutive words, whereas tw[j << s] of the flat table puts every
// lane of a wave on its own cache line from stage 1 on -- the texture-address unit, not the VALU, was the limit.
__global__ void gl_staged_twiddle_kernel(u64 *out, u64 w, u32 logn) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 n = 1ULL << logn;
    if (i >= n - 1) {
        if (i == n - 1) out[i] = 1;
        return;
    }
    // i = n - (n >> s) + j  with j < n >> (s + 1):  s = number of leading ones of i in logn bits
    u32 s = 0;
    while (i >= n - (n >> (s + 1))) s++;
    u64 j = i - (n - (n >> s));
    out[i] = gl_pow(w, j << s);
}

// ---------------------------------------------------------------- NTT passes
// One launch = k consecutive radix-2 stages of a size-2^logn transform, done in LDS on
// tiles of 2^(k+c+d) elements: all 2^k values of the k-bit index window the stages act on,
// times 2^c consecutive low indices (coalesced 8*2^c-byte runs), times 2^d high indices.
// Stage s (0-based from the top) pairs indices that differ in bit (logn-1-s) and uses the
// twiddle w^((i mod h) << s), h = 2^(logn-1-s), from the table tw[i] = w^i, i < 2^(logn-1).
//   DIF (forward order of stages): natural-order input -> bit-reversed output
//   DIT (reverse order of stages): bit-reversed input  -> natural-order output
struct ntt_pass {
    int logn, s0, k, c, d;
    int log_in;        // input indices >= 2^log_in read as zero (LDE zero padding); = logn otherwise
    int scale_shift;   // log2 of the low table size of the two-level load scale (0 = no load scale)
    u64 out_scale;     // multiplied into every stored value when != 1 (1/n of the inverse transform)
    // shift-twiddle kernel (gl_ntt_pass_g4_kernel): the k stages in ng groups of gsz[i] <= 4 stages (top stage first); goff[i] =
    // offset of the group's block in the table of per-element twiddles (gl_get_group_twiddles)
    int ng, gsz[4];
    u64 goff[4];
    int zskip_ok;      // the zero-aware first group of an 8x LDE may be used (off: ZKLC_NTT_ZSKIP=0)
};

template <bool DIT>
__global__ void __launch_bounds__(NTT_THREADS)
gl_ntt_pass_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t in_stride, size_t out_stride, ntt_pass p,
                   const u64 *__restrict__ tw, const u64 *__restrict__ scale_hi, const u64 *__restrict__ scale_lo) {
    __shared__ u64 tile[NTT_TILE_MAX];
    const int tile_log = p.k + p.c + p.d;
    const int tile_n = 1 << tile_log;
    const int lowbits = p.logn - p.s0 - p.k;
    const u32 tid = threadIdx.x;
    // tile coordinates
    const u64 t_id = blockIdx.x;
    const u64 tileL = t_id & ((1ULL << (lowbits - p.c)) - 1);
    const u64 tileH = t_id >> (lowbits - p.c);
    const u64 *src = in + (size_t)blockIdx.y * in_stride;
    u64 *dst = out + (size_t)blockIdx.y * out_stride;

    auto global_index = [&](u32 e) -> u64 {
        u64 lowc = e & ((1u << p.c) - 1);
        u64 mid = (e >> p.c) & ((1u << p.k) - 1);
        u64 hid = e >> (p.c + p.k);
        u64 L = (tileL << p.c) | lowc;
        u64 H = (tileH << p.d) | hid;
        return (H << (lowbits + p.k)) | (mid << lowbits) | L;
    };

    for (u32 e = tid; e < (u32)tile_n; e += NTT_THREADS) {
        u64 g = global_index(e);
        u64 v = 0;
        if (g < (1ULL << p.log_in)) {
            v = src[g];
            if (p.scale_shift) v = gl_mul(v, gl_mul(scale_hi[g >> p.scale_shift], scale_lo[g & ((1u << p.scale_shift) - 1)]));
        }
        tile[e] = v;
    }
    __syncthreads();

    for (int tt = 0; tt < p.k; tt++) {
        const int t = DIT ? (p.k - 1 - tt) : tt;
        const int s = p.s0 + t;
        const int pb = p.c + (p.k - 1 - t);  // tile-index bit of the paired elements
        for (u32 b = tid; b < (u32)(tile_n >> 1); b += NTT_THREADS) {
            u32 e0 = ((b >> pb) << (pb + 1)) | (b & ((1u << pb) - 1));
            u32 e1 = e0 | (1u << pb);
            u64 lowc = e0 & ((1u << p.c) - 1);
            u64 mid = (e0 >> p.c) & ((1u << p.k) - 1);
            u64 j = ((mid & ((1u << (p.k - 1 - t)) - 1)) << lowbits) | ((tileL << p.c) | lowc);
            u64 w = tw[j << s];
            u64 u = tile[e0], v = tile[e1];
            if (DIT) {
                v = gl_mul(v, w);
                tile[e0] = gl_add(u, v);
                tile[e1] = gl_sub(u, v);
            } else {
                tile[e0] = gl_add(u, v);
                tile[e1] = gl_mul(gl_sub(u, v), w);
            }
        }
        __syncthreads();
    }

    for (u32 e = tid; e < (u32)tile_n; e += NTT_THREADS) {
        u64 v = tile[e];
        if (p.out_scale != 1) v = gl_mul(v, p.out_scale);
        dst[global_index(e)] = v;
    }
}

// ---- radix-8 variant of the pass: the k stages of a pass are done in groups of 3 (then 2 or 1): a lane loads the 8
// elements of a radix-8 butterfly group from the LDS tile, does its 12 butterflies in registers and writes them back --
// a third of the LDS traffic and of the barriers of the radix-2 loop above, and 7 instead of 12 twiddle loads.
// For the group of stages s'..s'+g-1 (s' = s0 + t0) on elements base | m << pb_low the twiddle of the butterfly of stage
// s'+u whose lower element has group bits m is  w^(((m mod 2^(g-1-u)) << (logn - g + u)) + (J << (s' + u)))  with
// J = the index bits below the group; all of them are entries of the one table tw[i] = w^i, i < n/2.
// LDS tile padded by one element per 8 (index e + e/8) so that both the unit-stride and the stride-8 groups spread
// over the banks.
#define NTT_TI(e) ((e) + ((e) >> 3))
template <int G, bool DIT>
ZKLC_D void gl_ntt_group(u64 *tile, const u64 *__restrict__ tws, u64 n_full, u32 base, int pb_low, u64 J, int sh, int hi_shift) {
    constexpr int M = 1 << G;
    u64 x[M];
#pragma unroll
    for (int m = 0; m < M; m++) x[m] = tile[NTT_TI(base | ((u32)m << pb_low))];
#pragma unroll
    for (int uu = 0; uu < G; uu++) {
        const int u = DIT ? (G - 1 - uu) : uu;   // stage within the group; DIT runs the stages backwards
        const int bit = G - 1 - u;
#pragma unroll
        for (int m = 0; m < M; m++) {
            if (m & (1 << bit)) continue;
            const u64 lp = (u64)(m & ((1 << bit) - 1));
            u64 w = tws[(n_full - (n_full >> (sh + u))) + ((lp << hi_shift) + J)];
            u64 a = x[m], b = x[m | (1 << bit)];
            if (DIT) {
                b = gl_mul(b, w);
                x[m] = gl_add(a, b);
                x[m | (1 << bit)] = gl_sub(a, b);
            } else {
                x[m] = gl_add(a, b);
                x[m | (1 << bit)] = gl_mul(gl_sub(a, b), w);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < M; m++) tile[NTT_TI(base | ((u32)m << pb_low))] = x[m];
}

template <bool DIT>
__global__ void __launch_bounds__(NTT_THREADS_MAX)
gl_ntt_pass_r8_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t in_stride, size_t out_stride, ntt_pass p,
                      const u64 *__restrict__ tw, const u64 *__restrict__ scale_hi, const u64 *__restrict__ scale_lo) {
    extern __shared__ u64 tile[];      // 2^(k+c+d) elements + one pad per 8 (NTT_TI)
    const u32 n_threads = blockDim.x;
    const int tile_log = p.k + p.c + p.d;
    const int tile_n = 1 << tile_log;
    const int lowbits = p.logn - p.s0 - p.k;
    const u32 tid = threadIdx.x;
    const u64 t_id = blockIdx.x;
    const u64 tileL = t_id & ((1ULL << (lowbits - p.c)) - 1);
    const u64 tileH = t_id >> (lowbits - p.c);
    const u64 *src = in + (size_t)blockIdx.y * in_stride;
    u64 *dst = out + (size_t)blockIdx.y * out_stride;

    auto global_index = [&](u32 e) -> u64 {
        u64 lowc = e & ((1u << p.c) - 1);
        u64 mid = (e >> p.c) & ((1u << p.k) - 1);
        u64 hid = e >> (p.c + p.k);
        u64 L = (tileL << p.c) | lowc;
        u64 H = (tileH << p.d) | hid;
        return (H << (lowbits + p.k)) | (mid << lowbits) | L;
    };

    for (u32 e = tid; e < (u32)tile_n; e += n_threads) {
        u64 g = global_index(e);
        u64 v = 0;
        if (g < (1ULL << p.log_in)) {
            v = src[g];
            if (p.scale_shift) v = gl_mul(v, gl_mul(scale_hi[g >> p.scale_shift], scale_lo[g & ((1u << p.scale_shift) - 1)]));
        }
        tile[NTT_TI(e)] = v;
    }
    __syncthreads();

    int done = 0;
    while (done < p.k) {
        const int g = p.k - done >= 3 ? 3 : p.k - done;
        const int t0 = DIT ? (p.k - done - g) : done;      // first (lowest-numbered) stage of the group
        const int mg = p.k - g - t0;                        // number of `mid` bits below the group
        const int pb_low = p.c + mg;
        const int sh = p.s0 + t0;
        const int hi_shift = mg + lowbits;
        const u32 n_groups = (u32)tile_n >> g;
        for (u32 gi = tid; gi < n_groups; gi += n_threads) {
            u32 base = ((gi >> pb_low) << (pb_low + g)) | (gi & ((1u << pb_low) - 1));
            u64 lowc = base & ((1u << p.c) - 1);
            u64 mid_low = (base >> p.c) & ((1u << mg) - 1);
            u64 J = (mid_low << lowbits) | ((tileL << p.c) | lowc);
            if (g == 3)
                gl_ntt_group<3, DIT>(tile, tw, 1ULL << p.logn, base, pb_low, J, sh, hi_shift);
            else if (g == 2)
                gl_ntt_group<2, DIT>(tile, tw, 1ULL << p.logn, base, pb_low, J, sh, hi_shift);
            else
                gl_ntt_group<1, DIT>(tile, tw, 1ULL << p.logn, base, pb_low, J, sh, hi_shift);
        }
        __syncthreads();
        done += g;
    }

    for (u32 e = tid; e < (u32)tile_n; e += n_threads) {
        u64 v = tile[NTT_TI(e)];
        if (p.out_scale != 1) v = gl_mul(v, p.out_scale);
        dst[global_index(e)] = v;
    }
}

// ---- shift-twiddle variant (the product path; goldilocks_ntt_group.cuh): groups of up to FOUR stages, a lane holds the 16
// elements of a group; the twiddles inside a group are powers of two (shifts), the rest of every butterfly's twiddle is collected
// into one multiplication per element by an entry of the group's table block  tab[goff + (m - 1) * NJ + J] = w^((bitrev_g(m) J) << s').
// 15 general multiplications + 17 shifts per 32 butterflies instead of 32 multiplications, and ceil(k / 4) LDS round trips per
// pass.  LDS tile padded by one element per 16 (unit-stride and stride-16 / stride-8 groups all spread over the banks).
#define NTT_TJ(e) ((e) + ((e) >> 4))
template <int G, bool DIT, bool INV, int ZP = 0>
ZKLC_D void gl_ntt_group_lds(u64 *tile, const u64 *__restrict__ tabJ, u64 nj, u32 base, int pb_low) {
    constexpr int M = 1 << G;
    u64 x[M], t[M - 1 > 0 ? M - 1 : 1];
#pragma unroll
    for (int m = 1; m < M; m++) t[m - 1] = tabJ[(u64)(m - 1) * nj];
    // NTT_TJ is additive over disjoint bit fields: the lane's part once, the group-index part is wave-uniform (scalar)
    const u32 tjb = NTT_TJ(base);
    // ZP: only the first 2^(G - ZP) elements of the group exist (zero padding of an LDE): the others are neither read nor were
    // they written by the tile load
#pragma unroll
    for (int m = 0; m < (M >> ZP); m++) x[m] = tile[tjb + NTT_TJ((u32)m << pb_low)];
#pragma unroll
    for (int m = (M >> ZP); m < M; m++) x[m] = 0;
    gl_ntt_group_regs<G, DIT, INV, ZP>(x, t);
#pragma unroll
    for (int m = 0; m < M; m++) tile[tjb + NTT_TJ((u32)m << pb_low)] = x[m];
}

template <bool DIT, bool INV>
__global__ void __launch_bounds__(NTT_THREADS_MAX)
gl_ntt_pass_g4_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t in_stride, size_t out_stride, ntt_pass p,
                      const u64 *__restrict__ tab, const u64 *__restrict__ scale_hi, const u64 *__restrict__ scale_lo) {
    extern __shared__ u64 tile[];      // 2^(k+c+d) elements + one pad per 16 (NTT_TJ)
    const u32 n_threads = blockDim.x;
    const int tile_log = p.k + p.c + p.d;
    const int tile_n = 1 << tile_log;
    const int lowbits = p.logn - p.s0 - p.k;
    const u32 tid = threadIdx.x;
    const u64 t_id = blockIdx.x;
    const u64 tileL = t_id & ((1ULL << (lowbits - p.c)) - 1);
    const u64 tileH = t_id >> (lowbits - p.c);
    const u64 *src = in + (size_t)blockIdx.y * in_stride;
    u64 *dst = out + (size_t)blockIdx.y * out_stride;

    auto global_index = [&](u32 e) -> u64 {
        u64 lowc = e & ((1u << p.c) - 1);
        u64 mid = (e >> p.c) & ((1u << p.k) - 1);
        u64 hid = e >> (p.c + p.k);
        u64 L = (tileL << p.c) | lowc;
        u64 H = (tileH << p.d) | hid;
        return (H << (lowbits + p.k)) | (mid << lowbits) | L;
    };

    // tile load, eight elements of a lane at a time with all eight global loads in flight (one load -> wait -> LDS write per
    // iteration exposed the HBM latency sixteen times per tile: 39 % of the wave cycles were SQ_WAIT_ANY).  The zero padding of an
    // LDE (7 of 8 elements of its first pass) issues no load.
    // Index arithmetic (round 4): element e = tid | hi with hi a multiple of the (power-of-two) workgroup size; global_index scatters
    // disjoint bit fields of e to disjoint bit fields of the global index and NTT_TJ is additive over them, so the lane's part is
    // computed ONCE (g_tid, lds_tid) and the `hi` part of every element is wave-uniform scalar work: one OR per element instead of
    // re-deriving three bit fields (~12 % of the pass's VALU instructions went into these index computations).
    const u64 g_tid = global_index(tid);
    const u32 lds_tid = NTT_TJ(tid);
    const bool lane_in = tid < (u32)tile_n;               // tiles smaller than the workgroup (small transforms)
    // Zero padding of an 8x LDE (round 6): the first pass of a DIF transform sees its window bits on top, so the padded positions
    // are the elements e >= tile_n / 8 of every tile.  When the first group spans all three padded bits it runs in its
    // zero-aware form (goldilocks_ntt_group.cuh, ZP = 3): the padded seven eighths of the tile are neither loaded, scaled (two
    // multiplications per element that produced zeros) nor written to LDS -- the first group writes every position before the
    // second group reads any.  ZKLC_NTT_ZSKIP=0: the general form (A/B).
    bool zskip = false;
    if constexpr (!DIT) zskip = p.s0 == 0 && p.d == 0 && p.logn - p.log_in == 3 && p.gsz[0] >= 3 && p.zskip_ok;
    const u32 load_n = zskip ? (u32)tile_n >> 3 : (u32)tile_n;
    {
        constexpr int LD = 8;
        const u64 lim = 1ULL << p.log_in;
        const bool lane_ld = tid < load_n;
        for (u32 hi0 = 0; hi0 < load_n; hi0 += LD * n_threads) {
            u64 g[LD], v[LD], sh[LD], sl[LD];
#pragma unroll
            for (int q = 0; q < LD; q++) {
                const u32 hi = hi0 + q * n_threads;                      // wave-uniform
                g[q] = hi < load_n ? (g_tid | global_index(hi)) : g_tid;
                v[q] = 0;
                if (lane_ld && g[q] < lim) v[q] = src[g[q]];          // predicated, still all in flight: nothing below waits before the batch is issued
            }
            if (p.scale_shift) {
#pragma unroll
                for (int q = 0; q < LD; q++) {
                    const u64 gc = g[q] < lim ? g[q] : lim - 1;
                    sh[q] = scale_hi[gc >> p.scale_shift];
                    sl[q] = scale_lo[gc & ((1u << p.scale_shift) - 1)];
                }
            }
#pragma unroll
            for (int q = 0; q < LD; q++) {
                const u32 hi = hi0 + q * n_threads;
                if (hi >= load_n || !lane_ld) continue;
                u64 x = g[q] < lim ? v[q] : 0;
                if (p.scale_shift) x = gl_mul(x, gl_mul(sh[q], sl[q]));
                tile[lds_tid + NTT_TJ(hi)] = x;
            }
        }
    }
    __syncthreads();

    for (int gg = 0; gg < p.ng; gg++) {
        const int gi_ = DIT ? (p.ng - 1 - gg) : gg;         // DIT runs the groups (and the stages inside them) backwards
        const int g = p.gsz[gi_];
        int t0 = 0;                                          // first (lowest-numbered) stage of the group
        for (int q = 0; q < gi_; q++) t0 += p.gsz[q];
        const int mg = p.k - g - t0;                         // number of `mid` bits below the group
        const int pb_low = p.c + mg;
        const u64 nj = 1ULL << (mg + lowbits);
        const u64 *gtab = tab + p.goff[gi_];
        const u32 n_groups = (u32)tile_n >> g;
        for (u32 gi = tid; gi < n_groups; gi += n_threads) {
            u32 base = ((gi >> pb_low) << (pb_low + g)) | (gi & ((1u << pb_low) - 1));
            u64 lowc = base & ((1u << p.c) - 1);
            u64 mid_low = (base >> p.c) & ((1u << mg) - 1);
            u64 J = (mid_low << lowbits) | ((tileL << p.c) | lowc);
            if constexpr (!DIT) {
                if (zskip && gi_ == 0) {               // wave-uniform
                    if (g == 4)
                        gl_ntt_group_lds<4, DIT, INV, 3>(tile, gtab + J, nj, base, pb_low);
                    else
                        gl_ntt_group_lds<3, DIT, INV, 3>(tile, gtab + J, nj, base, pb_low);
                    continue;
                }
            }
            if (g == 4)
                gl_ntt_group_lds<4, DIT, INV>(tile, gtab + J, nj, base, pb_low);
            else if (g == 3)
                gl_ntt_group_lds<3, DIT, INV>(tile, gtab + J, nj, base, pb_low);
            else if (g == 2)
                gl_ntt_group_lds<2, DIT, INV>(tile, gtab + J, nj, base, pb_low);
            else
                gl_ntt_group_lds<1, DIT, INV>(tile, gtab + J, nj, base, pb_low);
        }
        __syncthreads();
    }

    {
        constexpr int LD = 8;
        for (u32 hi0 = 0; hi0 < (u32)tile_n; hi0 += LD * n_threads) {
            u64 v[LD];
#pragma unroll
            for (int q = 0; q < LD; q++) {
                const u32 hi = hi0 + q * n_threads;
                v[q] = tile[(lane_in ? lds_tid : 0u) + (hi < (u32)tile_n ? NTT_TJ(hi) : 0u)];
            }
#pragma unroll
            for (int q = 0; q < LD; q++) {
                const u32 hi = hi0 + q * n_threads;
                if (hi >= (u32)tile_n || !lane_in) continue;
                dst[g_tid | global_index(hi)] = p.out_scale != 1 ? gl_mul(v[q], p.out_scale) : v[q];
            }
        }
    }
}

// block of one group in the table of per-element twiddles: out[(m - 1) * nj + J] = w^((bitrev_g(m) * J) << s_first), 1 <= m < 2^g
__global__ void gl_group_twiddle_kernel(u64 *out, u64 w, u32 g, u32 s_first, u64 nj) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((1ULL << g) - 1) * nj) return;
    u32 m = 1 + (u32)(i / nj);
    u64 J = i % nj;
    out[i] = gl_pow(w, ((u64)gl_bitrev_small(m, (int)g) * J) << s_first);
}

// bit reversal of every polynomial of a batch
// in place: element i and element bitrev(i) change places (the lower index of every pair does the swap).  Round 5: replaces a
// device-to-device copy into scratch followed by the out-of-place kernel -- half the traffic, no scratch buffer.
__global__ void __launch_bounds__(256) gl_bitrev_inplace_kernel(u64 *__restrict__ data, size_t stride, int logn) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1ULL << logn)) return;
    u64 r = __brevll(i) >> (64 - logn);
    if (i < r) {
        u64 *p = data + (size_t)blockIdx.y * stride;
        u64 a = p[i], b = p[r];
        p[i] = b;
        p[r] = a;
    }
}

// ---------------------------------------------------------------- Poseidon / Merkle
__global__ void __launch_bounds__(256) poseidon_gl_permute_kernel(u64 *states, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
    u64 *p = states + (size_t)i * 12;
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = p[k];
    poseidon_gl_permute(s);
#pragma unroll
    for (int k = 0; k < 12; k++) p[k] = s[k];
}

// leaf i = (mat[p * stride + i * leaf_stride])_{p < width}; one lane per leaf.  Poly-major matrices (leaf_stride 1)
// are coalesced across lanes for every p; leaf_stride = width, stride = 1 reads row-major leaves (FRI commit trees).
__global__ void __launch_bounds__(256)
gl_hash_leaves_kernel(const u64 *__restrict__ mat, size_t stride, size_t leaf_stride, u32 width, u32 n_leaves,
                      u64 *__restrict__ digests) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    u64 h[4];
    poseidon_gl_hash_or_noop(mat + (size_t)i * leaf_stride, stride, width, h);
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(digests + (size_t)i * 4);
    o[0] = make_ulonglong2(h[0], h[1]);
    o[1] = make_ulonglong2(h[2], h[3]);
}

// parents[i] = two_to_one(children[2i], children[2i+1])
__global__ void __launch_bounds__(256) gl_merkle_level_kernel(const u64 *__restrict__ children, u64 *__restrict__ parents, u32 n_parents) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents) return;
    const ulonglong2 *c = reinterpret_cast<const ulonglong2 *>(children + (size_t)i * 8);
    ulonglong2 a = c[0], b = c[1], cc = c[2], dd = c[3];
    u64 l[4] = {a.x, a.y, b.x, b.y}, r[4] = {cc.x, cc.y, dd.x, dd.y}, h[4];
    poseidon_gl_two_to_one(l, r, h);
    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(parents + (size_t)i * 4);
    o[0] = make_ulonglong2(h[0], h[1]);
    o[1] = make_ulonglong2(h[2], h[3]);
}

// ---- cooperative Poseidon for SMALL trees: one permutation per 16-lane group (lane g < 12 holds state element g).
// A lane-per-permutation kernel needs ~35 k dependent instructions = ~58 us per Merkle level however few nodes the level has;
// the top ~13 levels of every tree and the small FRI trees are pure latency.  Here a permutation is ~5 k instructions per
// lane: S-box on the lane's own element, the MDS layer as 12 LDS reads of the group's elements (circulant row of lane g), and
// the partial rounds in the dense form -- S-box on lane 0 only, the same MDS layer, optimised round constants (first layer 12,
// then one per round) -- which is the same permutation (oracle/poseidon_gl.py::permute_naive, tests/test_oracle_poseidon.py).
#define PGL_COOP_GROUPS 16   // 16-lane groups per 256-thread workgroup
__device__ __forceinline__ u64 pgl_coop_mds(u64 s, u32 g, u64 *grp) {
    const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    // A 16-lane group lives inside ONE wave and owns its 12 LDS words: the LDS queue of a wave is in order, so the exchange needs no
    // workgroup barrier (round 5: two s_barrier per layer x 30 layers were most of a cooperative permutation's latency) -- only the
    // compiler must keep the order: previous layer's reads, this layer's writes, this layer's reads.
    // INVARIANT (every caller; checked where the groups are formed): blockDim.x is a multiple of 16, the 16 lanes of a group are
    // threadIdx.x >> 4 == const -- so a group never spans two waves (64 % 16 == 0) -- and `grp` is PRIVATE to the group.  A caller
    // whose groups crossed waves or shared `grp` would need __syncthreads() here instead.
    static_assert(64 % 16 == 0 && (PGL_COOP_GROUPS * 16) % 64 == 0, "a 16-lane group must live inside one wave");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // order for the compiler ...
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // ... and the previous layer's reads have returned
    if (g < 12) grp[g] = s;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the group's twelve words are in LDS before anyone reads them
    __builtin_amdgcn_wave_barrier();
    u64 sl = 0, sh = 0;
    const u32 gg = g < 12 ? g : 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u32 j = gg + i;
        j = j >= 12 ? j - 12 : j;
        u64 v = grp[j];
        sl += (u64)(u32)v * C[i];
        sh += (v >> 32) * C[i];
    }
    if (gg == 0) {
        sl += (u64)(u32)s * 8;
        sh += (s >> 32) * 8;
    }
    u64 l = sl + (sh << 32);
    u64 h = (sh >> 32) + (l < sl);
    return gl_reduce128_loose(l, h);
}
// permutes the group's state in place; all 16 lanes of the group must call it (wave-local exchange through `grp`)
__device__ __forceinline__ u64 pgl_coop_permute(u64 s, u32 g, u64 *grp) {
    const u32 gg = g < 12 ? g : 0;
#pragma unroll 1
    for (int r = 0; r < 4; r++) s = pgl_coop_mds(pgl_sbox_l(gl_add_lc(s, PGL_RC[12 * r + gg])), g, grp);
    s = gl_add_lc(s, PGL_FP_FIRST[gg]);
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        u64 t = pgl_sbox_l(s);
        if (r < 21) t = gl_add_lc(t, PGL_FP_RC[r]);
        s = pgl_coop_mds(g == 0 ? t : s, g, grp);
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) s = pgl_coop_mds(pgl_sbox_l(gl_add_lc(s, PGL_RC[12 * (26 + r) + gg])), g, grp);
    return gl_canonical(s);
}
// parents[i] = two_to_one(children[2i], children[2i+1]), one 16-lane group per parent
__global__ void __launch_bounds__(256)
gl_merkle_level_coop_kernel(const u64 *__restrict__ children, u64 *__restrict__ parents, u32 n_parents) {
    __shared__ u64 sh[PGL_COOP_GROUPS][12];
    const u32 g = threadIdx.x & 15, grp = threadIdx.x >> 4;
    u32 node = blockIdx.x * PGL_COOP_GROUPS + grp;
    const bool live = node < n_parents;
    if (!live) node = n_parents - 1;
    u64 s = g < 8 ? children[(size_t)node * 8 + g] : 0;
    s = pgl_coop_permute(s, g, sh[grp]);
    if (live && g < 4) parents[(size_t)node * 4 + g] = s;
}
// Up to FIVE levels in one launch (round 5): a workgroup owns 32 consecutive digests of level L and hashes them down -- 16 parents,
// 8, 4, 2, 1 -- writing every level where the per-level kernels would (the openings read them).  The levels of a tree are stored
// one after the other (zklc_gl_merkle_tree_words), level L + j at `lvl[j]`; `nlev` <= 5 levels are produced.  Between two levels
// the digests cross waves through LDS (one barrier per level); inside a permutation the exchange is wave-local.
struct gl_subtree_levels {
    u64 *lvl[6];          // lvl[0] = the children (read), lvl[1..nlev] = the parents' levels (written)
};
__global__ void __launch_bounds__(256)
gl_merkle_subtree_coop_kernel(gl_subtree_levels L, u32 n_children, u32 nlev) {
    __shared__ u64 sh[PGL_COOP_GROUPS][12];
    __shared__ u64 dig[32][4];
    const u32 g = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const u32 base = blockIdx.x * 32;                       // first child of this workgroup
    if (threadIdx.x < 128) {
        u32 c = base + (threadIdx.x >> 2);
        dig[threadIdx.x >> 2][threadIdx.x & 3] = c < n_children ? L.lvl[0][(size_t)c * 4 + (threadIdx.x & 3)] : 0;
    }
    __syncthreads();
    u32 width = 16;                                         // parents of this workgroup at the current level
#pragma unroll 1
    for (u32 j = 1; j <= nlev; j++, width >>= 1) {
        const bool live = grp < width;
        const u32 node = live ? grp : 0;
        u64 s = g < 8 ? dig[2 * node + (g >> 2)][g & 3] : 0;
        s = pgl_coop_permute(s, g, sh[grp]);
        __syncthreads();                                    // every group has read its two children
        const u32 gnode = (base >> j) + node;
        if (live && g < 4) {
            dig[node][g] = s;
            if (gnode < (n_children >> j)) L.lvl[j][(size_t)gnode * 4 + g] = s;
        }
        __syncthreads();
    }
}
// leaf digests (hash_or_noop of `width` elements), one 16-lane group per leaf; same addressing as gl_hash_leaves_kernel
__global__ void __launch_bounds__(256)
gl_hash_leaves_coop_kernel(const u64 *__restrict__ mat, size_t stride, size_t leaf_stride, u32 width, u32 n_leaves,
                           u64 *__restrict__ digests) {
    __shared__ u64 sh[PGL_COOP_GROUPS][12];
    const u32 g = threadIdx.x & 15, grp = threadIdx.x >> 4;
    u32 leaf = blockIdx.x * PGL_COOP_GROUPS + grp;
    const bool live = leaf < n_leaves;
    if (!live) leaf = n_leaves - 1;
    const u64 *in = mat + (size_t)leaf * leaf_stride;
    u64 s = 0;
    if (width <= 4) {
        if (g < width) s = in[(size_t)g * stride];
    } else {
#pragma unroll 1
        for (u32 off = 0; off < width; off += 8) {
            if (g < 8 && off + g < width) s = in[(size_t)(off + g) * stride];
            s = pgl_coop_permute(s, g, sh[grp]);
        }
    }
    if (live && g < 4) digests[(size_t)leaf * 4 + g] = s;
}
#define PGL_COOP_MAX_NODES (1u << 14)   // above this the lane-per-permutation kernels fill the chip and are cheaper per permutation

// ---------------------------------------------------------------- host side
static u64 host_gl_mul(u64 a, u64 b) {
    unsigned __int128 x = (unsigned __int128)a * b;
    return (u64)(x % GL_P);
}
static u64 host_gl_pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = host_gl_mul(r, a);
        a = host_gl_mul(a, a);
        e >>= 1;
    }
    return r;
}

void zklc_gl_fini(zklc_ctx *ctx) {
    for (int i = 0; i <= ZKLC_GL_MAX_LOG; i++) {
        if (ctx->gl_tw_fwd[i]) (void)hipFree(ctx->gl_tw_fwd[i]);
        if (ctx->gl_tw_inv[i]) (void)hipFree(ctx->gl_tw_inv[i]);
        if (ctx->gl_tws_fwd[i]) (void)hipFree(ctx->gl_tws_fwd[i]);
        if (ctx->gl_tws_inv[i]) (void)hipFree(ctx->gl_tws_inv[i]);
        if (ctx->gl_twg_fwd[i]) (void)hipFree(ctx->gl_twg_fwd[i]);
        if (ctx->gl_twg_inv[i]) (void)hipFree(ctx->gl_twg_inv[i]);
        ctx->gl_tw_fwd[i] = ctx->gl_tw_inv[i] = ctx->gl_tws_fwd[i] = ctx->gl_tws_inv[i] = nullptr;
        ctx->gl_twg_fwd[i] = ctx->gl_twg_inv[i] = nullptr;
    }
    if (ctx->gl_scale_hi) (void)hipFree(ctx->gl_scale_hi);
    if (ctx->gl_scale_lo) (void)hipFree(ctx->gl_scale_lo);
    ctx->gl_scale_hi = ctx->gl_scale_lo = nullptr;
}

// twiddle tables w^i (i < n/2) for the forward and inverse transforms of size 2^logn
static int32_t gl_get_twiddles(zklc_ctx *ctx, hipStream_t st, int logn, bool inverse, const u64 **out) {
    void **slot = inverse ? &ctx->gl_tw_inv[logn] : &ctx->gl_tw_fwd[logn];
    if (!*slot) {
        u64 n_half = logn ? (1ULL << (logn - 1)) : 1;
        ZKLC_HIP(ctx, hipMalloc(slot, n_half * 8));
        u64 w = host_gl_pow(GL_POWER_OF_TWO_GENERATOR, 1ULL << (32 - logn));
        if (inverse) w = host_gl_pow(w, GL_P - 2);
        hipLaunchKernelGGL(gl_pow_table_kernel, dim3((unsigned)((n_half + 255) / 256)), dim3(256), 0, st, (u64 *)*slot, w, n_half,
                           (u64)1);
        ZKLC_HIP(ctx, hipGetLastError());
    }
    *out = (const u64 *)*slot;
    return ZKLC_OK;
}

static int32_t gl_get_staged_twiddles(zklc_ctx *ctx, hipStream_t st, int logn, bool inverse, const u64 **out) {
    void **slot = inverse ? &ctx->gl_tws_inv[logn] : &ctx->gl_tws_fwd[logn];
    if (!*slot) {
        u64 n = 1ULL << logn;
        ZKLC_HIP(ctx, hipMalloc(slot, n * 8));
        u64 w = host_gl_pow(GL_POWER_OF_TWO_GENERATOR, 1ULL << (32 - logn));
        if (inverse) w = host_gl_pow(w, GL_P - 2);
        hipLaunchKernelGGL(gl_staged_twiddle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (u64 *)*slot, w, (u32)logn);
        ZKLC_HIP(ctx, hipGetLastError());
    }
    *out = (const u64 *)*slot;
    return ZKLC_OK;
}

#define GL_SCALE_LO_LOG 10
// two-level table of shift^j: hi[j >> 10] * lo[j & 1023]
static int32_t gl_get_scale(zklc_ctx *ctx, hipStream_t st, u64 shift, int logn) {
    int hi_n = logn > GL_SCALE_LO_LOG ? (1 << (logn - GL_SCALE_LO_LOG)) : 1;
    if (ctx->gl_scale_shift == shift && ctx->gl_scale_hi_n >= hi_n) return ZKLC_OK;
    if (ctx->gl_scale_hi) (void)hipFree(ctx->gl_scale_hi);
    ctx->gl_scale_hi = nullptr;
    if (!ctx->gl_scale_lo) ZKLC_HIP(ctx, hipMalloc(&ctx->gl_scale_lo, (1 << GL_SCALE_LO_LOG) * 8));
    int cap = hi_n < (1 << 14) ? (1 << 14) : hi_n;
    ZKLC_HIP(ctx, hipMalloc(&ctx->gl_scale_hi, (size_t)cap * 8));
    hipLaunchKernelGGL(gl_pow_table_kernel, dim3((1 << GL_SCALE_LO_LOG) / 256), dim3(256), 0, st, (u64 *)ctx->gl_scale_lo, shift,
                       (u64)(1 << GL_SCALE_LO_LOG), (u64)1);
    hipLaunchKernelGGL(gl_pow_table_kernel, dim3((cap + 255) / 256), dim3(256), 0, st, (u64 *)ctx->gl_scale_hi, shift, (u64)cap,
                       (u64)(1 << GL_SCALE_LO_LOG));
    ZKLC_HIP(ctx, hipGetLastError());
    ctx->gl_scale_shift = shift;
    ctx->gl_scale_hi_n = cap;
    return ZKLC_OK;
}

// Runs all passes of one transform.  dit=false: natural in -> bit-reversed out; dit=true: bit-reversed in -> natural out.
static int32_t gl_ntt_run(zklc_ctx *ctx, hipStream_t st, const u64 *in, size_t in_stride, u64 *out, size_t out_stride, int logn,
                          int log_in, u32 batch, bool inverse, bool dit, u64 load_shift) {
    if (logn == 0) {
        if (in != out)
            for (u32 b = 0; b < batch; b++)
                ZKLC_HIP(ctx, hipMemcpyAsync(out + b * out_stride, in + b * in_stride, 8, hipMemcpyDeviceToDevice, st));
        return ZKLC_OK;
    }
    static const bool radix2 = getenv("ZKLC_NTT_RADIX2") != nullptr;   // A/B switch: the original one-stage-per-barrier loop
    static const bool r8 = getenv("ZKLC_NTT_R8") != nullptr;           // A/B switch: radix-8 groups with a table twiddle per butterfly
    const bool lds_ok = zklc_once_per_device([] {     // 72 KB of dynamic LDS is above the default 64 KB cap
        const int bytes = ((1 << NTT_TILE_LOG_BIG) + (1 << NTT_TILE_LOG_BIG) / 8) * (int)sizeof(u64);
        const void *fns[6] = {(const void *)gl_ntt_pass_r8_kernel<true>,        (const void *)gl_ntt_pass_r8_kernel<false>,
                              (const void *)gl_ntt_pass_g4_kernel<true, true>,  (const void *)gl_ntt_pass_g4_kernel<true, false>,
                              (const void *)gl_ntt_pass_g4_kernel<false, true>, (const void *)gl_ntt_pass_g4_kernel<false, false>};
        for (const void *f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }) == hipSuccess;
    if (!lds_ok) return ZKLC_ERR_HIP;
    const bool g4 = !radix2 && !r8;
    const u64 *tw = nullptr;
    int32_t rc = ZKLC_OK;
    if (!g4) rc = radix2 ? gl_get_twiddles(ctx, st, logn, inverse, &tw) : gl_get_staged_twiddles(ctx, st, logn, inverse, &tw);
    if (rc) return rc;
    if (load_shift) {
        if ((rc = gl_get_scale(ctx, st, load_shift, log_in))) return rc;
    }
    // plan: strided windows from the top (each leaves >= 4 low bits in the tile: 128-byte runs), then one contiguous window.
    // Tile of 2^12 elements, or 2^13 where that saves a whole pass over the data (e.g. 2^21: two passes instead of three);
    // ZKLC_NTT_TILE=12|13 pins it (A/B).
    static const int tile_pin = getenv("ZKLC_NTT_TILE") ? atoi(getenv("ZKLC_NTT_TILE")) : 0;
    auto n_passes = [&](int T) { return 1 + (logn > T ? (logn - T + (T - 4) - 1) / (T - 4) : 0); };
    int T = NTT_TILE_LOG_MAX;
    if (!radix2 && (tile_pin == NTT_TILE_LOG_BIG || (tile_pin == 0 && n_passes(NTT_TILE_LOG_BIG) < n_passes(NTT_TILE_LOG_MAX))))
        T = NTT_TILE_LOG_BIG;
    int ks[8], np = 0;
    int k_last = logn < T ? logn : T;
    int remaining = logn - k_last;
    for (int m = n_passes(T) - 1; m > 0; m--) {      // even split of the strided bits
        int k = (remaining + m - 1) / m;
        ks[np++] = k;
        remaining -= k;
    }
    ks[np++] = k_last;
    u64 n_inv = inverse ? host_gl_pow(1ULL << logn, GL_P - 2) : 1;
    int s0_of[8], s0 = 0;
    for (int i = 0; i < np; i++) {
        s0_of[i] = s0;
        s0 += ks[i];
    }
    // groups of <= 4 stages per pass (sizes as even as possible: 13 = 4 + 3 + 3 + 3) and their blocks in the twiddle table; the
    // plan depends on logn only, so the table is built once per (logn, direction) and context
    int ng_of[8], gsz_of[8][4];
    u64 goff_of[8][4], tab_elems = 0;
    for (int i = 0; i < np; i++) {
        int ng = (ks[i] + 3) / 4, base = ks[i] / ng, extra = ks[i] % ng, t0 = 0;
        ng_of[i] = ng;
        for (int j = 0; j < ng; j++) {
            int g = base + (j < extra ? 1 : 0);
            gsz_of[i][j] = g;
            goff_of[i][j] = tab_elems;
            tab_elems += ((1ULL << g) - 1) << (logn - (s0_of[i] + t0) - g);
            t0 += g;
        }
    }
    if (g4) {
        // resident per (context, logn, direction); built on first use.  The build is serialised and COMPLETE before the pointer is
        // published (another host thread or stream of the same context must never see a half-built block)
        static std::mutex twg_mutex;
        std::lock_guard<std::mutex> twg_lock(twg_mutex);
        void **slot = inverse ? &ctx->gl_twg_inv[logn] : &ctx->gl_twg_fwd[logn];
        if (!*slot) {
            void *fresh = nullptr;
            ZKLC_HIP(ctx, hipMalloc(&fresh, tab_elems * 8));
            void **build = &fresh;
            u64 w = host_gl_pow(GL_POWER_OF_TWO_GENERATOR, 1ULL << (32 - logn));
            if (inverse) w = host_gl_pow(w, GL_P - 2);
            for (int i = 0; i < np; i++) {
                int t0 = 0;
                for (int j = 0; j < ng_of[i]; j++) {
                    int g = gsz_of[i][j], sf = s0_of[i] + t0;
                    u64 nj = 1ULL << (logn - sf - g), cnt = ((1ULL << g) - 1) * nj;
                    hipLaunchKernelGGL(gl_group_twiddle_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st,
                                       (u64 *)*build + goff_of[i][j], w, (u32)g, (u32)sf, nj);
                    t0 += g;
                }
            }
            ZKLC_HIP(ctx, hipGetLastError());
            ZKLC_HIP(ctx, zklc_stream_wait(st));       // once per (logn, direction)
            *slot = fresh;
        }
        tw = (const u64 *)*slot;
    }
    for (int ii = 0; ii < np; ii++) {
        int i = dit ? (np - 1 - ii) : ii;  // DIT runs the windows bottom-up
        ntt_pass p;
        p.logn = logn;
        p.s0 = s0_of[i];
        p.k = ks[i];
        int lowbits = logn - p.s0 - p.k;
        p.c = lowbits < (T - p.k) ? lowbits : (T - p.k);
        int room = T - p.k - p.c;
        p.d = p.s0 < room ? p.s0 : room;
        bool first = (ii == 0), last = (ii == np - 1);
        p.log_in = first ? log_in : logn;
        p.scale_shift = (first && load_shift) ? GL_SCALE_LO_LOG : 0;
        p.out_scale = last ? n_inv : 1;
        static const bool zskip_on = !(getenv("ZKLC_NTT_ZSKIP") && getenv("ZKLC_NTT_ZSKIP")[0] == '0');
        p.zskip_ok = zskip_on ? 1 : 0;
        p.ng = ng_of[i];
        for (int j = 0; j < 4; j++) {
            p.gsz[j] = j < ng_of[i] ? gsz_of[i][j] : 0;
            p.goff[j] = j < ng_of[i] ? goff_of[i][j] : 0;
        }
        const u64 *src = first ? in : out;
        size_t sstride = first ? in_stride : out_stride;
        dim3 grid((unsigned)(1ULL << (logn - (p.k + p.c + p.d))), batch);
        if (g4) {
            const int tile_log = p.k + p.c + p.d;
            const unsigned threads = tile_log > NTT_TILE_LOG_MAX ? NTT_THREADS_MAX : NTT_THREADS;
            const size_t lds = ((size_t(1) << tile_log) + (size_t(1) << tile_log) / 16) * sizeof(u64);
            const u64 *sh = (const u64 *)ctx->gl_scale_hi, *sl = (const u64 *)ctx->gl_scale_lo;
            if (dit && inverse)
                hipLaunchKernelGGL((gl_ntt_pass_g4_kernel<true, true>), grid, dim3(threads), lds, st, src, out, sstride, out_stride, p, tw, sh, sl);
            else if (dit)
                hipLaunchKernelGGL((gl_ntt_pass_g4_kernel<true, false>), grid, dim3(threads), lds, st, src, out, sstride, out_stride, p, tw, sh, sl);
            else if (inverse)
                hipLaunchKernelGGL((gl_ntt_pass_g4_kernel<false, true>), grid, dim3(threads), lds, st, src, out, sstride, out_stride, p, tw, sh, sl);
            else
                hipLaunchKernelGGL((gl_ntt_pass_g4_kernel<false, false>), grid, dim3(threads), lds, st, src, out, sstride, out_stride, p, tw, sh, sl);
        } else if (radix2) {
            if (dit)
                hipLaunchKernelGGL(gl_ntt_pass_kernel<true>, grid, dim3(NTT_THREADS), 0, st, src, out, sstride, out_stride, p, tw,
                                   (const u64 *)ctx->gl_scale_hi, (const u64 *)ctx->gl_scale_lo);
            else
                hipLaunchKernelGGL(gl_ntt_pass_kernel<false>, grid, dim3(NTT_THREADS), 0, st, src, out, sstride, out_stride, p, tw,
                                   (const u64 *)ctx->gl_scale_hi, (const u64 *)ctx->gl_scale_lo);
        } else {
            const int tile_log = p.k + p.c + p.d;
            const unsigned threads = tile_log > NTT_TILE_LOG_MAX ? NTT_THREADS_MAX : NTT_THREADS;
            const size_t lds = ((size_t(1) << tile_log) + (size_t(1) << tile_log) / 8) * sizeof(u64);
            if (dit)
                hipLaunchKernelGGL(gl_ntt_pass_r8_kernel<true>, grid, dim3(threads), lds, st, src, out, sstride, out_stride, p, tw,
                                   (const u64 *)ctx->gl_scale_hi, (const u64 *)ctx->gl_scale_lo);
            else
                hipLaunchKernelGGL(gl_ntt_pass_r8_kernel<false>, grid, dim3(threads), lds, st, src, out, sstride, out_stride, p, tw,
                                   (const u64 *)ctx->gl_scale_hi, (const u64 *)ctx->gl_scale_lo);
        }
        ZKLC_HIP(ctx, hipGetLastError());
    }
    return ZKLC_OK;
}

extern "C" int32_t zklc_gl_ntt_dev(zklc_ctx *ctx, void *stream, uint64_t *d_data, uint32_t log_n, uint32_t batch,
                                   uint32_t flags, uint64_t coset_shift) {
    if (!ctx || !d_data || log_n > ZKLC_GL_MAX_LOG) return ZKLC_ERR_INVALID_ARG;
    if (batch == 0) return ZKLC_OK;
    bool inverse = flags & ZKLC_NTT_INVERSE, in_br = flags & ZKLC_NTT_IN_BITREV, out_br = flags & ZKLC_NTT_OUT_BITREV;
    if (in_br && out_br) return ZKLC_ERR_INVALID_ARG;
    if (coset_shift && (inverse || in_br)) return ZKLC_ERR_INVALID_ARG;  // coset scaling is applied to natural-order coefficients
    if (coset_shift >= GL_P) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = zklc_pick_stream(ctx, stream);
    size_t n = 1ULL << log_n;
    if (in_br) return gl_ntt_run(ctx, st, d_data, n, d_data, n, log_n, log_n, batch, inverse, true, 0);
    int32_t rc = gl_ntt_run(ctx, st, d_data, n, d_data, n, log_n, log_n, batch, inverse, false, coset_shift);
    if (rc || out_br) return rc;
    // natural order requested: swap i <-> bitrev(i) in place
    hipLaunchKernelGGL(gl_bitrev_inplace_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, st, d_data, n, (int)log_n);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

int32_t zklc_gl_intt_copy_dev(zklc_ctx *ctx, hipStream_t st, const uint64_t *d_src, uint64_t *d_dst, uint32_t log_n, uint32_t batch) {
    if (!ctx || !d_src || !d_dst || log_n > ZKLC_GL_MAX_LOG) return ZKLC_ERR_INVALID_ARG;
    if (batch == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t n = 1ULL << log_n;
    int32_t rc = gl_ntt_run(ctx, st, d_src, n, d_dst, n, log_n, log_n, batch, true, false, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(gl_bitrev_inplace_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, st, d_dst, n, (int)log_n);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" int32_t zklc_gl_lde_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_coeffs, uint32_t log_n, uint32_t rate_bits,
                                   uint32_t batch, uint64_t coset_shift, uint64_t *d_out, uint32_t flags) {
    if (!ctx || !d_coeffs || !d_out || log_n + rate_bits > ZKLC_GL_MAX_LOG || coset_shift >= GL_P) return ZKLC_ERR_INVALID_ARG;
    if (batch == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = zklc_pick_stream(ctx, stream);
    int logN = log_n + rate_bits;
    size_t n = 1ULL << log_n, N = 1ULL << logN;
    int32_t rc = gl_ntt_run(ctx, st, d_coeffs, n, d_out, N, logN, log_n, batch, false, false, coset_shift);
    if (rc || (flags & ZKLC_NTT_OUT_BITREV)) return rc;
    hipLaunchKernelGGL(gl_bitrev_inplace_kernel, dim3((unsigned)((N + 255) / 256), batch), dim3(256), 0, st, d_out, N, logN);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" int32_t zklc_poseidon_gl_permute_dev(zklc_ctx *ctx, void *stream, uint64_t *d_states, uint32_t n) {
    if (!ctx || (n && !d_states)) return ZKLC_ERR_INVALID_ARG;
    if (n == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(poseidon_gl_permute_kernel, dim3((n + 255) / 256), dim3(256), 0, zklc_pick_stream(ctx, stream), d_states, n);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" uint64_t zklc_gl_merkle_tree_words(uint32_t log_leaves, uint32_t cap_height) {
    if (cap_height > log_leaves) return 0;
    uint64_t words = 0;
    for (uint32_t l = 0; l <= log_leaves - cap_height; l++) words += 4ULL << (log_leaves - l);
    return words;
}

extern "C" int32_t zklc_gl_merkle_commit_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_mat, uint64_t stride, uint32_t log_leaves,
                                             uint32_t width, uint32_t cap_height, uint64_t *d_tree) {
    if (!ctx || !d_mat || !d_tree || log_leaves > 30 || cap_height > log_leaves || width == 0) return ZKLC_ERR_INVALID_ARG;
    if (stride < (1ULL << log_leaves)) return ZKLC_ERR_INVALID_ARG;
    return zklc_gl_merkle_commit_strided(ctx, zklc_pick_stream(ctx, stream), d_mat, stride, 1, log_leaves, width, cap_height, d_tree);
}

int32_t zklc_gl_merkle_commit_strided(zklc_ctx *ctx, hipStream_t st, const uint64_t *d_mat, uint64_t stride, uint64_t leaf_stride,
                                      uint32_t log_leaves, uint32_t width, uint32_t cap_height, uint64_t *d_tree) {
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    u32 n = 1u << log_leaves;
    static const bool coop = !getenv("ZKLC_NO_COOP_POSEIDON");
    if (coop && n <= PGL_COOP_MAX_NODES)
        hipLaunchKernelGGL(gl_hash_leaves_coop_kernel, dim3((n + PGL_COOP_GROUPS - 1) / PGL_COOP_GROUPS), dim3(256), 0, st, d_mat,
                           (size_t)stride, (size_t)leaf_stride, width, n, d_tree);
    else
        hipLaunchKernelGGL(gl_hash_leaves_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_mat, (size_t)stride, (size_t)leaf_stride,
                           width, n, d_tree);
    ZKLC_HIP(ctx, hipGetLastError());
    // ZKLC_MERKLE_FUSED=0 keeps one launch per level (A/B and the variant parity test)
    static const bool fused = !(getenv("ZKLC_MERKLE_FUSED") && getenv("ZKLC_MERKLE_FUSED")[0] == '0');
    u64 *level = d_tree;
    const u32 total = log_leaves - cap_height;
    for (u32 l = 0; l < total;) {
        u32 parents = n >> (l + 1);
        u64 *next = level + (4ULL << (log_leaves - l));
        if (coop && fused && parents <= PGL_COOP_MAX_NODES && parents >= 16 && total - l >= 2) {
            // the cooperative levels in groups of up to five per launch: a workgroup hashes 32 digests down to one
            u32 nlev = total - l < 5 ? total - l : 5;
            while (nlev > 1 && ((2 * parents) >> nlev) == 0) nlev--;
            gl_subtree_levels L;
            u64 *p = level;
            for (u32 j = 0; j <= nlev; j++) {
                L.lvl[j] = p;
                p += 4ULL << (log_leaves - l - j);
            }
            for (u32 j = nlev + 1; j < 6; j++) L.lvl[j] = nullptr;
            hipLaunchKernelGGL(gl_merkle_subtree_coop_kernel, dim3((2 * parents + 31) / 32), dim3(256), 0, st, L, 2 * parents, nlev);
            ZKLC_HIP(ctx, hipGetLastError());
            level = L.lvl[nlev];
            l += nlev;
            continue;
        }
        if (coop && parents <= PGL_COOP_MAX_NODES)
            hipLaunchKernelGGL(gl_merkle_level_coop_kernel, dim3((parents + PGL_COOP_GROUPS - 1) / PGL_COOP_GROUPS), dim3(256), 0, st,
                               (const u64 *)level, next, parents);
        else
            hipLaunchKernelGGL(gl_merkle_level_kernel, dim3((parents + 255) / 256), dim3(256), 0, st, (const u64 *)level, next, parents);
        ZKLC_HIP(ctx, hipGetLastError());
        level = next;
        l++;
    }
    return ZKLC_OK;
}

// ---- host-pointer flavours (stage through the context buffers) ----
extern "C" int32_t zklc_gl_ntt(zklc_ctx *ctx, uint64_t *data, uint32_t log_n, uint32_t batch, uint32_t flags, uint64_t coset_shift) {
    if (!ctx || !data || log_n > ZKLC_GL_MAX_LOG) return ZKLC_ERR_INVALID_ARG;
    if (batch == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t bytes = ((size_t)batch << log_n) * 8;
    void *d;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, bytes, &d))) return rc;
    ZKLC_HIP(ctx, hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = zklc_gl_ntt_dev(ctx, ctx->stream, (uint64_t *)d, log_n, batch, flags, coset_shift))) return rc;
    ZKLC_HIP(ctx, zklc_readback_async(data, d, bytes, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}

extern "C" int32_t zklc_gl_lde(zklc_ctx *ctx, const uint64_t *coeffs, uint32_t log_n, uint32_t rate_bits, uint32_t batch,
                               uint64_t coset_shift, uint64_t *out, uint32_t flags) {
    if (!ctx || !coeffs || !out || log_n + rate_bits > ZKLC_GL_MAX_LOG) return ZKLC_ERR_INVALID_ARG;
    if (batch == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t in_bytes = ((size_t)batch << log_n) * 8, out_bytes = ((size_t)batch << (log_n + rate_bits)) * 8;
    void *din, *dout;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, in_bytes, &din))) return rc;
    if ((rc = zklc_stage(ctx, 1, out_bytes, &dout))) return rc;
    ZKLC_HIP(ctx, hipMemcpyAsync(din, coeffs, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = zklc_gl_lde_dev(ctx, ctx->stream, (const uint64_t *)din, log_n, rate_bits, batch, coset_shift, (uint64_t *)dout, flags)))
        return rc;
    ZKLC_HIP(ctx, zklc_readback_async(out, dout, out_bytes, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}

extern "C" int32_t zklc_poseidon_gl_permute(zklc_ctx *ctx, uint64_t *states, uint32_t n) {
    if (!ctx || (n && !states)) return ZKLC_ERR_INVALID_ARG;
    if (n == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t bytes = (size_t)n * 96;
    void *d;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, bytes, &d))) return rc;
    ZKLC_HIP(ctx, hipMemcpyAsync(d, states, bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = zklc_poseidon_gl_permute_dev(ctx, ctx->stream, (uint64_t *)d, n))) return rc;
    ZKLC_HIP(ctx, zklc_readback_async(states, d, bytes, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}

extern "C" int32_t zklc_gl_merkle_commit(zklc_ctx *ctx, const uint64_t *mat, uint64_t stride, uint32_t log_leaves, uint32_t width,
                                         uint32_t cap_height, uint64_t *tree_out) {
    if (!ctx || !mat || !tree_out || log_leaves > 30 || cap_height > log_leaves || width == 0) return ZKLC_ERR_INVALID_ARG;
    if (stride < (1ULL << log_leaves)) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t in_bytes = (size_t)stride * width * 8, tree_bytes = zklc_gl_merkle_tree_words(log_leaves, cap_height) * 8;
    void *dm, *dt;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, in_bytes, &dm))) return rc;
    if ((rc = zklc_stage(ctx, 1, tree_bytes, &dt))) return rc;
    ZKLC_HIP(ctx, hipMemcpyAsync(dm, mat, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = zklc_gl_merkle_commit_dev(ctx, ctx->stream, (const uint64_t *)dm, stride, log_leaves, width, cap_height, (uint64_t *)dt)))
        return rc;
    ZKLC_HIP(ctx, zklc_readback_async(tree_out, dt, tree_bytes, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}
