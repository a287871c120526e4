// plonky2 prover on gfx950: Z / partial products, quotient (gate constraints), openings, FRI, proof-of-work,
// query openings -- everything of `prove_with_partition_witness` after the witness is known.
//
// Replaces the un-vendored plonky2-near@2244a9d `plonk/prover.rs`, `plonk/vanishing_poly.rs`, `fri/oracle.rs`,
// `fri/prover.rs` behind near_bft_finality/src/prove_crypto/ed25519.rs:60,100 and recursion.rs:95.  The verifier
// side of every formula used here is restated in the reference tree: gnark-plonky2-verifier/plonk/plonk.go:60-250
// (vanishing polynomial), fri/fri.go:187-497 (domain order, combine, fold, final polynomial),
// challenger/challenger.go:42-166 (transcript).
//
// Layout in HBM: every polynomial batch is POLY-MAJOR; LDE matrices are in bit-reversed index order (position p holds
// the evaluation at g * w_N^bitrev(p)), which is the order of the Merkle leaves, so the commit kernels read them
// directly and a wave of 64 consecutive points reads 512 contiguous bytes of every column.  FRI oracles are arrays of
// extension elements (2 x u64), i.e. row-major leaves of 2 * arity words.  The Fiat-Shamir transcript (a few hundred
// Poseidon permutations) runs on the host between kernels (plonky2_host.cpp).
#include <algorithm>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <new>
#include <vector>
#include <string.h>
#include "plonky2_gates.cuh"
#include "plonky2_host.h"
#include "zklc_internal.h"

#define P2_THREADS 256

struct p2_challenges {
    u64 beta[P2_MAX_CH], gamma[P2_MAX_CH], alpha[P2_MAX_CH];
};

// ------------------------------------------------------------------------------------------------ small kernels
__global__ void p2_pow_table_kernel(u64 *out, u64 base, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl_pow(base, i);
}

// data[b][j] *= base^j
__global__ void p2_scale_by_powers_kernel(u64 *data, u64 base, u64 n) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    u64 *p = data + (size_t)blockIdx.y * n + j;
    *p = gl_mul(*p, gl_pow(base, j));
}

// out[j] = z^j (extension)
__global__ void p2_ext_powers_kernel(gl2 *out, gl2 z, u64 n) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = gl2_pow(z, j);
}

// out[12 j ..] = the 22-bit limbs (gl_limbs22) of the two coordinates of z^j: the constant table of gl_acc3_mul
__global__ void p2_ext_pow_limbs_kernel(u32 *out, gl2 z, u32 n) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    gl2 v = gl2_pow(z, j);
    gl_limbs22(v.a, out + 12 * (size_t)j);
    gl_limbs22(v.b, out + 12 * (size_t)j + 6);
}

// ---------------------------------------------------------------------------------------- Z and partial products
// prover.rs `wires_permutation_partial_products_and_zs`: per row the running products of the chunk quotients
//   rp[k][i] = prod_{k' <= k} prod_{j in chunk k'} (w_j + beta k_j x_i + gamma) / (w_j + beta sigma_j(x_i) + gamma)
__global__ void __launch_bounds__(P2_THREADS)
p2_chunk_products_kernel(const u64 *__restrict__ wires, const u64 *__restrict__ sigmas, const u64 *__restrict__ subgroup,
                         const u64 *__restrict__ k_is, u32 n, u32 routed, u32 qdf, u32 nchunks, u64 beta, u64 gamma,
                         u64 *__restrict__ rp) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 x = subgroup[i];
    u64 nums[16], dens[16];
    for (u32 k = 0; k < nchunks; k++) {
        u64 np = 1, dp = 1;
        u32 end = (k + 1) * qdf < routed ? (k + 1) * qdf : routed;
        for (u32 j = k * qdf; j < end; j++) {
            u64 w = gl_add(wires[(size_t)j * n + i], gamma);
            np = gl_mul(np, gl_add(w, gl_mul(beta, gl_mul(k_is[j], x))));
            dp = gl_mul(dp, gl_add(w, gl_mul(beta, sigmas[(size_t)j * n + i])));
        }
        nums[k] = np;
        dens[k] = dp;
    }
    // one inversion for all chunk denominators of the row
    u64 pref[16];
    u64 acc = 1;
    for (u32 k = 0; k < nchunks; k++) {
        pref[k] = acc;
        acc = gl_mul(acc, dens[k]);
    }
    u64 inv = gl_inv(acc);
    for (u32 k = nchunks; k-- > 0;) {
        u64 dinv = gl_mul(inv, pref[k]);
        inv = gl_mul(inv, dens[k]);
        nums[k] = gl_mul(nums[k], dinv);
    }
    acc = 1;
    for (u32 k = 0; k < nchunks; k++) {
        acc = gl_mul(acc, nums[k]);
        rp[(size_t)k * n + i] = acc;
    }
}

// exclusive prefix PRODUCT over r[0..n): phase 1 = per-block (1024 elements) local scan + block totals
#define P2_SCAN_ITEMS 4
#define P2_SCAN_BLOCK (P2_THREADS * P2_SCAN_ITEMS)
__global__ void __launch_bounds__(P2_THREADS) p2_scan_local_kernel(const u64 *__restrict__ r, u64 *__restrict__ excl,
                                                                     u64 *__restrict__ totals, u32 n) {
    __shared__ u64 sh[P2_THREADS];
    u32 base = blockIdx.x * P2_SCAN_BLOCK + threadIdx.x * P2_SCAN_ITEMS;
    u64 v[P2_SCAN_ITEMS], acc = 1;
#pragma unroll
    for (int k = 0; k < P2_SCAN_ITEMS; k++) {
        v[k] = base + k < n ? r[base + k] : 1;
        acc = gl_mul(acc, v[k]);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 off = 1; off < P2_THREADS; off <<= 1) {
        u64 t = threadIdx.x >= off ? sh[threadIdx.x - off] : 1;
        __syncthreads();
        sh[threadIdx.x] = gl_mul(sh[threadIdx.x], t);
        __syncthreads();
    }
    u64 pre = threadIdx.x ? sh[threadIdx.x - 1] : 1;
#pragma unroll
    for (int k = 0; k < P2_SCAN_ITEMS; k++) {
        if (base + k < n) excl[base + k] = pre;
        pre = gl_mul(pre, v[k]);
    }
    if (threadIdx.x == P2_THREADS - 1) totals[blockIdx.x] = sh[P2_THREADS - 1];
}
// phase 2: exclusive scan of the block totals (single block, serial per thread over a strip)
__global__ void __launch_bounds__(P2_THREADS) p2_scan_totals_kernel(u64 *totals, u32 nblocks, u64 *grand_total) {
    __shared__ u64 sh[P2_THREADS];
    u32 per = (nblocks + P2_THREADS - 1) / P2_THREADS;
    u32 s = threadIdx.x * per, e = s + per < nblocks ? s + per : nblocks;
    u64 acc = 1;
    for (u32 i = s; i < e; i++) acc = gl_mul(acc, totals[i]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 off = 1; off < P2_THREADS; off <<= 1) {
        u64 t = threadIdx.x >= off ? sh[threadIdx.x - off] : 1;
        __syncthreads();
        sh[threadIdx.x] = gl_mul(sh[threadIdx.x], t);
        __syncthreads();
    }
    u64 pre = threadIdx.x ? sh[threadIdx.x - 1] : 1;
    for (u32 i = s; i < e; i++) {
        u64 t = totals[i];
        totals[i] = pre;
        pre = gl_mul(pre, t);
    }
    if (threadIdx.x == P2_THREADS - 1) *grand_total = sh[P2_THREADS - 1];
}
// phase 3: Z[i] = block prefix * local prefix; partial product k = Z[i] * rp[k][i]
__global__ void __launch_bounds__(P2_THREADS)
p2_z_apply_kernel(const u64 *__restrict__ excl, const u64 *__restrict__ totals, const u64 *__restrict__ rp, u32 n, u32 npp,
                  u64 *__restrict__ z_out, u64 *__restrict__ pp_out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = gl_mul(excl[i], totals[i / P2_SCAN_BLOCK]);
    z_out[i] = z;
    for (u32 k = 0; k < npp; k++) pp_out[(size_t)k * n + i] = gl_mul(z, rp[(size_t)k * n + i]);
}

// ---------------------------------------------------------------------------------------------------- quotient
struct p2_quotient_args {
    const u64 *cs, *wires, *zs;     // LDE matrices (bit-reversed), stride N
    const p2_gate *gates;
    const u64 *extra, *k_is;
    u32 lde_bits, degree_bits, rate_bits, num_constants, nsel, routed, nch, npp, qdf, num_gates;
    u64 w_lde;                      // primitive 2^lde_bits-th root of unity
    u64 zh_inv[16], zh[16];         // 1 / (x^n - 1) and x^n - 1 for the 2^rate_bits cosets of <w_n> inside g<w_N>
    u64 n_field;                    // n as a field element
    u64 pih[4];
    p2_challenges ch;
    const u32 *apow[P2_MAX_CH];     // powers of the alphas as 22-bit limbs (gl_limbs22: 6 words per power), one table per challenge
    u64 *out;                       // [nch][N]
    const u64 *xs, *l0;             // per LDE point (bit-reversed order): x = g w^bitrev(p) and L_0(x) = (x^n - 1) / (n (x - 1))
    u32 num_wires;
};

// vanishing_poly.rs `eval_vanishing_poly_base_batch` at the point stored at position p, divided by Z_H.
// The sum over the vanishing terms is split over launches so that each has a small register footprint (the all-in-one
// kernel needed 221 VGPRs = 2 waves/SIMD): p2_quotient_base_kernel writes the L_0 and partial-product terms,
// p2_quotient_gate_kernel<TYPE> adds filter_g * sum_i alpha^(k_gates + i) c_{g,i} for one gate of the list.
__global__ void __launch_bounds__(P2_THREADS) p2_quotient_base_kernel(p2_quotient_args a) {
    const size_t N = (size_t)1 << a.lde_bits;
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    u64 i = __brevll((u64)p) >> (64 - a.lde_bits);  // natural index on the coset
    u64 x = a.xs[p];
    const u32 coset = (u32)i & ((1u << a.rate_bits) - 1);
    u64 l0 = a.l0[p];
    u64 i_next = (i + (1ULL << a.rate_bits)) & (N - 1);
    size_t p_next = (size_t)(__brevll(i_next) >> (64 - a.lde_bits));

    p2_consumer out;
    out.nch = a.nch;
    for (int c = 0; c < P2_MAX_CH; c++) out.apow[c] = (gl_ktab *)a.apow[c];
    out.reset(0);
    // L_0(x) (Z(x) - 1)
    for (u32 c = 0; c < a.nch; c++) out.emit(gl_mul(l0, gl_sub(a.zs[(size_t)c * N + p], 1)));
    // partial products (plonk.go:84-119)
    const u32 nchunks = a.npp + 1;
    for (u32 c = 0; c < a.nch; c++) {
        u64 beta = a.ch.beta[c], gamma = a.ch.gamma[c];
        u64 prev = a.zs[(size_t)c * N + p];
        for (u32 k = 0; k < nchunks; k++) {
            u64 np = 1, dp = 1;
            u32 end = (k + 1) * a.qdf < a.routed ? (k + 1) * a.qdf : a.routed;
            gl_ktab64 *kis = (gl_ktab64 *)a.k_is;
            u32 j = k * a.qdf;
            if (end - j == 8) {          // the usual chunk (quotient degree factor 8): sixteen loads in flight
                u64 wv[8], sv[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    wv[q] = a.wires[(size_t)(j + q) * N + p];
                    sv[q] = a.cs[(size_t)(a.num_constants + j + q) * N + p];
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    u64 w = gl_add(wv[q], gamma);
                    np = gl_mul(np, gl_add(w, gl_mul(beta, gl_mul(kis[j + q], x))));
                    dp = gl_mul(dp, gl_add(w, gl_mul(beta, sv[q])));
                }
                j = end;
            }
            for (; j < end; j++) {
                u64 w = gl_add(a.wires[(size_t)j * N + p], gamma);
                np = gl_mul(np, gl_add(w, gl_mul(beta, gl_mul(kis[j], x))));
                dp = gl_mul(dp, gl_add(w, gl_mul(beta, a.cs[(size_t)(a.num_constants + j) * N + p])));
            }
            u64 next = k + 1 < nchunks ? a.zs[(size_t)(a.nch + c * a.npp + k) * N + p] : a.zs[(size_t)c * N + p_next];
            out.emit(gl_sub(gl_mul(prev, np), gl_mul(next, dp)));
            prev = next;
        }
    }
    const u64 zi = a.zh_inv[coset];
#pragma unroll
    for (int c = 0; c < P2_MAX_CH; c++)      // static indices: a dynamic one would push the accumulators into memory
        if (c < (int)a.nch) a.out[(size_t)c * N + p] = gl_mul(out.result(c), zi);
}

// gate constraints, filtered (evaluate_gates.go:59-105); TYPE is a compile-time constant so only that evaluator is inlined.
// One launch takes ALL gates of the list that have this type (the eight U32AddMany variants of the Ed25519 circuit, the two
// BaseSum variants): they read nearly the same wire columns, and the second to eighth pass over a wave's 64 points finds them in
// the memory-side cache instead of streaming the LDE matrix from HBM once per variant.
#define P2_GATE_LIST_MAX 8
struct p2_gate_list {
    u32 n, idx[P2_GATE_LIST_MAX];
};
template <int TYPE>
__global__ void __launch_bounds__(P2_THREADS) p2_quotient_gate_kernel(p2_quotient_args a, p2_gate_list list) {
    const size_t N = (size_t)1 << a.lde_bits;
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    u64 i = __brevll((u64)p) >> (64 - a.lde_bits);
    const u32 coset = (u32)i & ((1u << a.rate_bits) - 1);
    p2_vars v;
    v.wires = a.wires;
    v.consts = a.cs;
    v.stride = N;
    v.p = p;
    v.nsel = a.nsel;
    for (int k = 0; k < 4; k++) v.pih[k] = a.pih[k];
    p2_consumer out;
    out.nch = a.nch;
    for (int c = 0; c < P2_MAX_CH; c++) out.apow[c] = (gl_ktab *)a.apow[c];
    u64 sum[P2_MAX_CH];
    for (int c = 0; c < P2_MAX_CH; c++) sum[c] = 0;
#pragma unroll 1
    for (u32 t = 0; t < list.n; t++) {
        const u32 g = list.idx[t];
        p2_gate gate = a.gates[g];
        gate.type = TYPE;
        out.reset(a.nch + a.nch * (a.npp + 1));   // the gate constraints follow the Z1 and partial-product terms
        if constexpr (TYPE == P2_POSEIDON_LOOSE)
            p2_eval_poseidon_loose(v, out);
        else if constexpr (TYPE >= P2_POSEIDON_LAZY)
            p2_eval_poseidon_lazy<p2_vars, TYPE - P2_POSEIDON_LAZY>(v, out);   // only these instantiations carry the opt-in evaluators
        else
            p2_eval_gate(gate, v, a.extra, out);
        u64 f = p2_filter(g, gate.group_start, gate.group_end, v.sel(gate.selector_index), a.nsel > 1);
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)      // static indices: a dynamic one would push the accumulators into memory
            if (c < (int)a.nch) sum[c] = gl_add(sum[c], gl_mul(f, out.result(c)));
    }
    const u64 zi = a.zh_inv[coset];
#pragma unroll
    for (int c = 0; c < P2_MAX_CH; c++)
        if (c < (int)a.nch) {
            u64 *o = a.out + (size_t)c * N + p;
            *o = gl_add(*o, gl_mul(sum[c], zi));
        }
}

// ---- all U32AddMany gates of a circuit in ONE evaluator (the Ed25519 circuit has eight variants, 3..16 addends).
// Every variant keeps 16 + 2 two-bit limbs per operation after its routed wires, and the limb regions of the variants overlap
// (columns 54..233 of the 234): evaluated gate by gate that is 900 range checks x (x-1)(x-2)(x-3) and 1 476 column loads per LDE
// point -- 24.8 GB of HBM traffic for a 3.9 GB matrix (profiles/r03q_pmc_*), the kernel was bandwidth-bound.  The range check of
// a column does not depend on the variant: here a lane walks the limb columns ONCE, from the top (the Horner sums of every variant
// then run in step), computes the range product of a column once (four columns per asm batch) and adds it to the accumulator of
// every variant whose limb region holds the column, at that variant's constraint index; `comb - res` is added as two terms
// (alpha^k comb here, alpha^k (p - res) when the routed wires are read), the sums being linear.  180 range checks and 756 loads.
// Constraint indices as in p2_eval_u32_add_many: per operation i: 21 i = the sum, 21 i + 1 + (17 - l) = limb l, 21 i + 19 / + 20 =
// result / carry recombination.
// MEASURED (profiles/r03v_prove_ed25519_kernel_stats.csv): 5.49 ms, the same as the eight per-gate evaluations it replaces, with 60 %
// less traffic and 55 % fewer VALU instructions: the eight static copies of the per-variant step (the accumulators must be indexed
// statically to stay in registers) make the kernel ~80 KB of code -- more than the 64 KB instruction cache two CUs share -- and 45 %
// of its instructions are SALU index arithmetic (column -> operation / limb / constraint index per variant) with a scalar table load
// per constraint.  Bit-exact (the proofs of the real circuit verify, bytes equal the C prover's); kept as an opt-in experiment
// (ZKLC_P2_ADDMANY=multi), the per-gate launches stay the default.
#define P2_AM_MAX 8
// one variant's share of a four-column step of p2_quotient_addmany_multi_kernel (a template so that V is static: the eight copies
// are too large for `#pragma unroll`, and a rolled loop would index the accumulators dynamically = scratch)
template <int V>
__device__ __forceinline__ void p2_am_variant_step(gl_acc3 (&acc)[P2_AM_MAX][P2_MAX_CH], u64 (&comb)[P2_AM_MAX], u32 ops, u32 l0, u32 c0,
                                                   u32 lo, const u64 *w, const u64 *rp, const p2_quotient_args &a, u32 k0) {
    auto emit = [&](u32 krel, u64 val) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)
            if (c < (int)a.nch) gl_acc3_mul(acc[V][c], val, (gl_ktab *)a.apow[c] + 6 * (size_t)(k0 + krel));
    };
    const u32 rel0 = c0 - l0;                         // wraps below the region; ops = 0 for the unused slots
    const u32 i0 = rel0 / 18, l_top = rel0 - 18 * i0;
    if (rel0 < 18 * ops && l_top >= 3) {
        // the usual case: the four columns are limbs l_top .. l_top - 3 of ONE operation, i.e. four consecutive constraints:
        // the alpha powers of the four are adjacent in the table (one batch of scalar loads instead of four dependent ones)
        const u32 kb = 21 * i0 + 18 - l_top;
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)
            if (c < (int)a.nch) {
                gl_ktab *t = (gl_ktab *)a.apow[c] + 6 * (size_t)(k0 + kb);
#pragma unroll
                for (int q = 0; q < 4; q++) gl_acc3_mul(acc[V][c], rp[q], t + 6 * q);
            }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            comb[V] = p2_horner4(comb[V], w[q]);
            if (l_top - q == 16) {
                emit(21 * i0 + 20, gl_canonical(comb[V]));
                comb[V] = 0;
            } else if (l_top - q == 0) {
                emit(21 * i0 + 19, gl_canonical(comb[V]));
                comb[V] = 0;
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {                     // region edges and steps that straddle two operations
        if (c0 < lo + q) continue;
        const u32 rel = c0 - q - l0;
        if (rel >= 18 * ops) continue;
        const u32 i = rel / 18, l = rel - 18 * i;
        emit(21 * i + 1 + (17 - l), rp[q]);
        comb[V] = p2_horner4(comb[V], w[q]);
        if (l == 16) {
            emit(21 * i + 20, gl_canonical(comb[V]));
            comb[V] = 0;
        } else if (l == 0) {
            emit(21 * i + 19, gl_canonical(comb[V]));
            comb[V] = 0;
        }
    }
}
// (eight accumulator sets = 96 VGPRs: two waves per SIMD; left to its occupancy heuristic the compiler spills them to scratch)
__global__ void __launch_bounds__(P2_THREADS) __attribute__((amdgpu_waves_per_eu(1, 2)))
p2_quotient_addmany_multi_kernel(p2_quotient_args a, p2_gate_list list) {
    const size_t N = (size_t)1 << a.lde_bits;
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    u64 pi = __brevll((u64)p) >> (64 - a.lde_bits);
    const u32 coset = (u32)pi & ((1u << a.rate_bits) - 1);
    const u64 *W = a.wires + p;
    const u32 k0 = a.nch + a.nch * (a.npp + 1);      // the gate constraints follow the Z1 and partial-product terms
    gl_acc3 acc[P2_AM_MAX][P2_MAX_CH];
#pragma unroll
    for (int v = 0; v < P2_AM_MAX; v++)
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++) acc[v][c].c0 = acc[v][c].c1 = acc[v][c].c2 = 0;
    auto emit = [&](int v, u32 krel, u64 val) __attribute__((always_inline)) {      // not inlined at every site = `acc` in scratch
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)
            if (c < (int)a.nch) gl_acc3_mul(acc[v][c], val, (gl_ktab *)a.apow[c] + 6 * (size_t)(k0 + krel));
    };
    // ---- routed wires, variant by variant: sum constraint, and the -res / -carry halves of the recombination constraints
    u32 lo = 0xFFFFFFFFu, hi = 0;                     // the union of the limb regions
    u32 OPS[P2_AM_MAX], L0[P2_AM_MAX];                // wave-uniform: operations and first limb column of every variant
#pragma unroll
    for (int v = 0; v < P2_AM_MAX; v++) {
        OPS[v] = L0[v] = 0;
        if (v >= (int)list.n) continue;
        const p2_gate g = a.gates[list.idx[v]];
        const u32 na = g.p[0], ops = g.p[1], per = na + 3;
        const u32 l0 = per * ops;
        OPS[v] = ops;
        L0[v] = l0;
        lo = l0 < lo ? l0 : lo;
        hi = l0 + 18 * ops > hi ? l0 + 18 * ops : hi;
#pragma unroll 1
        for (u32 i = 0; i < ops; i++) {
            p2_vars pv;
            pv.wires = a.wires;
            pv.stride = N;
            pv.p = p;
            u64 rc[2];
            p2_load<2>(pv, per * i + na + 1, rc);
            u64 sum = p2_sum_wires(pv, per * i, na + 1);          // the addends and the carry in
            emit(v, 21 * i, gl_sub(gl_add(gl_mul(rc[1], 1ULL << 32), rc[0]), sum));
            emit(v, 21 * i + 19, gl_neg(rc[0]));
            emit(v, 21 * i + 20, gl_neg(rc[1]));
        }
    }
    // ---- limb columns hi-1 .. lo, four per step (the next four are in flight while these are evaluated)
    u64 comb[P2_AM_MAX];
#pragma unroll
    for (int v = 0; v < P2_AM_MAX; v++) comb[v] = 0;
    auto load4 = [&](u32 top, u64 *w) __attribute__((always_inline)) {               // w[q] = column top - q (0 below `lo`: no variant uses it)
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = top >= lo + q ? W[(size_t)(top - q) * N] : 0;
    };
    u64 w[4], nxt[4];
    if (hi > lo) load4(hi - 1, w);
#pragma unroll 1
    for (u32 top = hi; top > lo; top = top >= 4 ? top - 4 : 0) {
        const u32 c0 = top - 1;                       // the columns of this step: c0, c0 - 1, c0 - 2, c0 - 3
        if (c0 >= lo + 4) load4(c0 - 4, nxt);
        u64 rp[4];
        p2_range_products4<4>(w, rp);
#define P2_AM_STEP(V) p2_am_variant_step<V>(acc, comb, OPS[V], L0[V], c0, lo, w, rp, a, k0);
        P2_AM_STEP(0) P2_AM_STEP(1) P2_AM_STEP(2) P2_AM_STEP(3) P2_AM_STEP(4) P2_AM_STEP(5) P2_AM_STEP(6) P2_AM_STEP(7)
#undef P2_AM_STEP
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = nxt[q];
        if (top < 4) break;
    }
    u64 sum[P2_MAX_CH];
    for (int c = 0; c < P2_MAX_CH; c++) sum[c] = 0;
    p2_vars sv;
    sv.consts = a.cs;
    sv.stride = N;
    sv.p = p;
    sv.nsel = a.nsel;
#pragma unroll
    for (int v = 0; v < P2_AM_MAX; v++) {
        if (v >= (int)list.n) continue;
        const u32 gi = list.idx[v];
        const p2_gate g = a.gates[gi];
        u64 f = p2_filter(gi, g.group_start, g.group_end, sv.sel(g.selector_index), a.nsel > 1);
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)
            if (c < (int)a.nch) sum[c] = gl_add(sum[c], gl_mul(f, gl_acc3_reduce(acc[v][c])));
    }
    const u64 zi = a.zh_inv[coset];
#pragma unroll
    for (int c = 0; c < P2_MAX_CH; c++)
        if (c < (int)a.nch) {
            u64 *o = a.out + (size_t)c * N + p;
            *o = gl_add(*o, gl_mul(sum[c], zi));
        }
}

// ---- all U32AddMany variants over an LDS tile shared by the FOUR WAVES of a workgroup (round 4; the default for circuits with several
// variants).  What the two forms above leave on the table: evaluated gate by gate, the range product x (x-1)(x-2)(x-3) of a limb column
// is recomputed by every variant whose limb region holds it -- 900 products per LDE point for 180 distinct columns, ~45 k of the
// kernel's ~68 k lane-instructions per point, and the wire matrix is fetched 3.2 times (profiles/r03q_pmc_*); the one-pass kernel
// (p2_quotient_addmany_multi_kernel) computes each product once but needs eight static copies of the per-variant step in ONE wave
// (80 KB of code, SALU-bound).  Here a workgroup owns 64 LDE points: the limb columns are walked from the top in phases of
// P2_AMT_COLS; in a phase every wave loads and range-checks a quarter of the columns ONCE (twelve loads in flight, three asm batches)
// and leaves (wire, product) in LDS; after the barrier every wave consumes the tile for ITS OWN two variants (p2_amt_plan: balanced
// by operation count on the host) -- wave-uniform constraint indices, so the alpha powers stay scalar loads, two static copies of the
// step, accumulators of two variants in registers.  Every column is read from HBM once (the routed wires once more), every product
// computed once; 48 KB of LDS per workgroup = three workgroups per CU.  Same field values as the per-gate form (the sums commute).
#define P2_AMT_COLS 48u
#define P2_AMT_WAVES 4
struct p2_amt_plan {
    u32 slot[P2_AMT_WAVES][2];      // position in the gate list of the (up to) two variants of every wave; 0xFFFFFFFF = none
    u32 lo, hi;                     // the union of the limb regions: columns [lo, hi)
};
__global__ void __launch_bounds__(64 * P2_AMT_WAVES) p2_quotient_addmany_tile_kernel(p2_quotient_args a, p2_gate_list list, p2_amt_plan plan) {
    __shared__ u64 tw[P2_AMT_COLS * 64], trp[P2_AMT_COLS * 64];
    const size_t N = (size_t)1 << a.lde_bits;
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t p = (size_t)blockIdx.x * 64 + lane;
    const u64 *W = a.wires + p;
    const u32 k0 = a.nch + a.nch * (a.npp + 1);       // the gate constraints follow the Z1 and partial-product terms
    gl_acc3 acc[2][P2_MAX_CH];
    u64 comb[2] = {0, 0};
    gl_ktab *apow[P2_MAX_CH];
#pragma unroll
    for (int c = 0; c < P2_MAX_CH; c++) apow[c] = (gl_ktab *)a.apow[c];
    u32 OPS[2], L0[2], GI[2];                         // wave-uniform: this wave's variants
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++) acc[s][c].c0 = acc[s][c].c1 = acc[s][c].c2 = 0;
        OPS[s] = L0[s] = 0;
        GI[s] = 0xFFFFFFFFu;
        const u32 pos = plan.slot[wave][s];
        if (pos == 0xFFFFFFFFu) continue;
        GI[s] = list.idx[pos];
        const p2_gate g = a.gates[GI[s]];
        const u32 na = g.p[0], ops = g.p[1], per = na + 3;
        OPS[s] = ops;
        L0[s] = per * ops;
        // routed wires: the sum constraint and the -res / -carry halves of the two recombination constraints (linear: the limb
        // halves are added when the tile is consumed)
        p2_vars pv;
        pv.wires = a.wires;
        pv.stride = N;
        pv.p = p;
        if (s == 0)
            p2_amt_routed<0>(acc, pv, na, ops, apow, (int)a.nch, k0);
        else
            p2_amt_routed<1>(acc, pv, na, ops, apow, (int)a.nch, k0);
    }
    // ---- limb columns hi-1 .. lo in phases of P2_AMT_COLS.  Round 5: the columns of phase k + 1 are REQUESTED (global loads into
    // registers) right after the barrier that publishes phase k's tile, so their latency passes under the consumption of phase k
    // instead of in front of the next barrier (r04: SQ_WAIT_ANY 49 % of the wave cycles, VALU 58 % of the issue ceiling).
    const u32 per_wave = P2_AMT_COLS / P2_AMT_WAVES;  // 12 = three asm batches of four range products
    u64 wn[per_wave];
    u32 top = plan.hi, base = top - plan.lo > P2_AMT_COLS ? top - P2_AMT_COLS : plan.lo;
#pragma unroll
    for (u32 j = 0; j < per_wave; j++) {
        const u32 first = base + wave * per_wave;
        const u32 col = first + j < top ? first + j : top - 1;              // clamped: a duplicate of the last column, not stored
        wn[j] = W[(size_t)col * N];
    }
#pragma unroll 1
    while (true) {
        {
            u64 w[per_wave], rp[per_wave];
            const u32 first = base + wave * per_wave;
#pragma unroll
            for (u32 j = 0; j < per_wave; j++) w[j] = wn[j];
            p2_range_products4<per_wave>(w, rp);
#pragma unroll
            for (u32 j = 0; j < per_wave; j++)
                if (first + j < top) {
                    tw[(size_t)(first + j - base) * 64 + lane] = w[j];
                    trp[(size_t)(first + j - base) * 64 + lane] = rp[j];
                }
        }
        __syncthreads();
        const u32 ntop = base;
        const bool more = ntop > plan.lo;
        const u32 nbase = !more ? plan.lo : (ntop - plan.lo > P2_AMT_COLS ? ntop - P2_AMT_COLS : plan.lo);
        if (more) {
            const u32 first = nbase + wave * per_wave;
#pragma unroll
            for (u32 j = 0; j < per_wave; j++) {
                const u32 col = first + j < ntop ? first + j : ntop - 1;
                wn[j] = W[(size_t)col * N];
            }
        }
        if (OPS[0]) p2_amt_consume<0>(acc, comb, OPS[0], L0[0], base, top, tw, trp, lane, apow, (int)a.nch, k0);
        if (OPS[1]) p2_amt_consume<1>(acc, comb, OPS[1], L0[1], base, top, tw, trp, lane, apow, (int)a.nch, k0);
        __syncthreads();
        if (!more) break;
        top = ntop;
        base = nbase;
    }
    // ---- filter_v * sum_v per wave, the waves' sums through LDS, one read-modify-write of the output per point
    u64 sum[P2_MAX_CH];
#pragma unroll
    for (int c = 0; c < P2_MAX_CH; c++) sum[c] = 0;
    p2_vars sv;
    sv.consts = a.cs;
    sv.stride = N;
    sv.p = p;
    sv.nsel = a.nsel;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (GI[s] == 0xFFFFFFFFu) continue;
        const p2_gate g = a.gates[GI[s]];
        const u64 f = p2_filter(GI[s], g.group_start, g.group_end, sv.sel(g.selector_index), a.nsel > 1);
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)
            if (c < (int)a.nch) sum[c] = gl_add(sum[c], gl_mul(f, gl_acc3_reduce(acc[s][c])));
    }
    u64 *part = tw;                                   // [wave][challenge][lane]; the tile is free after the last barrier
#pragma unroll
    for (int c = 0; c < P2_MAX_CH; c++) part[((size_t)wave * P2_MAX_CH + c) * 64 + lane] = sum[c];
    __syncthreads();
    if (wave == 0) {
        const u64 pi = __brevll((u64)p) >> (64 - a.lde_bits);
        const u64 zi = a.zh_inv[(u32)pi & ((1u << a.rate_bits) - 1)];
#pragma unroll
        for (int c = 0; c < P2_MAX_CH; c++)
            if (c < (int)a.nch) {
                u64 t = sum[c];
                for (u32 w = 1; w < P2_AMT_WAVES; w++) t = gl_add(t, part[((size_t)w * P2_MAX_CH + c) * 64 + lane]);
                u64 *o = a.out + (size_t)c * N + p;
                *o = gl_add(*o, gl_mul(t, zi));
            }
    }
}
// host: variants -> waves, heaviest first onto the least loaded wave with a free slot.  false = the list does not fit the kernel
static bool p2_amt_make_plan(const p2_gate *gates, const p2_gate_list &list, p2_amt_plan &plan) {
    if (list.n < 2 || list.n > 2 * P2_AMT_WAVES) return false;
    u32 load[P2_AMT_WAVES] = {0, 0, 0, 0}, used[P2_AMT_WAVES] = {0, 0, 0, 0};
    for (int w = 0; w < P2_AMT_WAVES; w++) plan.slot[w][0] = plan.slot[w][1] = 0xFFFFFFFFu;
    plan.lo = 0xFFFFFFFFu;
    plan.hi = 0;
    bool taken[P2_GATE_LIST_MAX] = {};
    for (u32 r = 0; r < list.n; r++) {
        u32 best = 0xFFFFFFFFu;
        for (u32 t = 0; t < list.n; t++)
            if (!taken[t] && (best == 0xFFFFFFFFu || gates[list.idx[t]].p[1] > gates[list.idx[best]].p[1])) best = t;
        taken[best] = true;
        const p2_gate &g = gates[list.idx[best]];
        const u32 ops = g.p[1], l0 = (g.p[0] + 3) * ops;
        if (ops == 0 || 24 * ops >= 480) return false;      // accumulator head room (p2_consumer normalises every 480 products)
        u32 w = 0xFFFFFFFFu;
        for (u32 k = 0; k < P2_AMT_WAVES; k++)
            if (used[k] < 2 && (w == 0xFFFFFFFFu || load[k] < load[w])) w = k;
        plan.slot[w][used[w]++] = best;
        load[w] += ops;
        plan.lo = l0 < plan.lo ? l0 : plan.lo;
        plan.hi = l0 + 18 * ops > plan.hi ? l0 + 18 * ops : plan.hi;
    }
    return plan.hi > plan.lo;
}

typedef void (*p2_gate_kernel_fn)(p2_quotient_args, p2_gate_list);
static p2_gate_kernel_fn p2_gate_kernel_of(u32 type) {
    // ZKLC_P2_POSEIDON_GATE: the evaluator of the Poseidon gate.  Default (round 5) = `loose`: the plain evaluator's loop structure in
    // the loose arithmetic of the hash kernel's C++ permutation (2.80 -> 2.53 ms per Ed25519 proof); `plain` = the canonical-value
    // form that was the default until round 4; `lazy` / `lazy1` = the whole-round / lazy-partial-round forms (plonky2_gates.cuh).
    // All four give the same field values: tests/test_gpu_plonky2.py proves one circuit under each and compares the bytes.
    static const char *pg = getenv("ZKLC_P2_POSEIDON_GATE");
    if (type == P2_POSEIDON && (!pg || !pg[0] || !strcmp(pg, "loose"))) return p2_quotient_gate_kernel<P2_POSEIDON_LOOSE>;
    if (type == P2_POSEIDON && pg && !strncmp(pg, "lazy", 4)) {
        // lazy: statements + unrolled partial rounds; lazy1: the partial rounds as rolled loops over per-lane LDS arrays
        return pg[4] == '1' ? p2_quotient_gate_kernel<P2_POSEIDON_LAZY + 1> : p2_quotient_gate_kernel<P2_POSEIDON_LAZY>;
    }
    switch (type) {
#define P2_CASE(T) case T: return p2_quotient_gate_kernel<T>;
        P2_CASE(P2_CONSTANT) P2_CASE(P2_PUBLIC_INPUT) P2_CASE(P2_ARITHMETIC) P2_CASE(P2_ARITHMETIC_EXT) P2_CASE(P2_MUL_EXT)
        P2_CASE(P2_BASE_SUM) P2_CASE(P2_POSEIDON) P2_CASE(P2_POSEIDON_MDS) P2_CASE(P2_RANDOM_ACCESS) P2_CASE(P2_REDUCING)
        P2_CASE(P2_REDUCING_EXT) P2_CASE(P2_EXPONENTIATION) P2_CASE(P2_COSET_INTERPOLATION) P2_CASE(P2_U32_ARITHMETIC)
        P2_CASE(P2_U32_ADD_MANY) P2_CASE(P2_U32_SUBTRACTION) P2_CASE(P2_U32_RANGE_CHECK) P2_CASE(P2_COMPARISON)
        P2_CASE(P2_U32_INTERLEAVE) P2_CASE(P2_UNINTERLEAVE_TO_U32) P2_CASE(P2_UNINTERLEAVE_TO_B32)
#undef P2_CASE
        default: return nullptr;   // P2_NOOP: no constraints
    }
}

// The fused form.  The per-gate launches above read the wire matrix once per gate type (~16x the algorithmic bytes for the 20
// gate types of the Ed25519 circuit); here a workgroup stages the constants and wires of 64 consecutive LDE points in LDS once
// ([column][64 points]: a lane's access is one conflict-free ds_read_b64) and its waves share the tile: every wave evaluates
// its own sub-list of jobs (a job = one gate of the list, or the L_0 / partial-product terms) for the same 64 points, the
// per-wave sums are added through LDS.  The job lists are balanced on the host from the measured cost of each gate
// (p2_plan_quotient).  Same field values as the per-gate form (the sum over gates commutes), so the proof bytes do not change.
#define P2_FQ_MAX_WAVES 8
#define P2_FQ_MAX_JOBS 40
#define P2_JOB_BASE 0xFFFFu
struct p2_fused_plan {
    u32 nwaves;
    u32 njobs[P2_FQ_MAX_WAVES];
    uint16_t jobs[P2_FQ_MAX_WAVES][P2_FQ_MAX_JOBS];
};

__global__ void __launch_bounds__(64 * P2_FQ_MAX_WAVES) p2_quotient_fused_kernel(p2_quotient_args a, p2_fused_plan plan) {
    extern __shared__ u64 p2_tile[];   // [num_constants + num_wires][64], then [nwaves][P2_MAX_CH][64] partial sums
    const size_t N = (size_t)1 << a.lde_bits;
    const u32 lane = threadIdx.x & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t p = (size_t)blockIdx.x * 64 + lane;
    const u32 ncols = a.num_constants + a.num_wires;
    {   // stage the tile: wave w takes columns w, w + nwaves, ...; eight loads in flight per lane
        u32 j = wave;
        for (; j + 7 * plan.nwaves < ncols; j += 8 * plan.nwaves) {
            u64 t[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                u32 jj = j + q * plan.nwaves;
                const u64 *src = jj < a.num_constants ? a.cs + (size_t)jj * N : a.wires + (size_t)(jj - a.num_constants) * N;
                t[q] = src[p];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) p2_tile[(size_t)(j + q * plan.nwaves) * 64 + lane] = t[q];
        }
        for (; j < ncols; j += plan.nwaves) {
            const u64 *src = j < a.num_constants ? a.cs + (size_t)j * N : a.wires + (size_t)(j - a.num_constants) * N;
            p2_tile[(size_t)j * 64 + lane] = src[p];
        }
    }
    __syncthreads();
    p2_vars_lds v;
    v.consts = (p2_lds_u64 *)p2_tile;
    v.wires = v.consts + (size_t)a.num_constants * 64;
    v.lane = lane;
    v.nsel = a.nsel;
    for (int k = 0; k < 4; k++) v.pih[k] = a.pih[k];
    u64 sum[P2_MAX_CH] = {0, 0};
    p2_consumer out;
    out.nch = a.nch;
    for (int c = 0; c < P2_MAX_CH; c++) out.apow[c] = (gl_ktab *)a.apow[c];
    const u32 njobs = plan.njobs[wave];
    for (u32 jb = 0; jb < njobs; jb++) {
        const u32 g = plan.jobs[wave][jb];
        if (g == P2_JOB_BASE) {
            u64 i = __brevll((u64)p) >> (64 - a.lde_bits);
            u64 x = a.xs[p], l0 = a.l0[p];
            u64 i_next = (i + (1ULL << a.rate_bits)) & (N - 1);
            size_t p_next = (size_t)(__brevll(i_next) >> (64 - a.lde_bits));
            out.reset(0);
            for (u32 c = 0; c < a.nch; c++) out.emit(gl_mul(l0, gl_sub(a.zs[(size_t)c * N + p], 1)));
            const u32 nchunks = a.npp + 1;
            for (u32 c = 0; c < a.nch; c++) {
                u64 beta = a.ch.beta[c], gamma = a.ch.gamma[c];
                u64 prev = a.zs[(size_t)c * N + p];
                for (u32 k = 0; k < nchunks; k++) {
                    u64 np = 1, dp = 1;
                    u32 end = (k + 1) * a.qdf < a.routed ? (k + 1) * a.qdf : a.routed;
                    for (u32 j = k * a.qdf; j < end; j++) {
                        u64 w = gl_add(v.w(j), gamma);
                        np = gl_mul(np, gl_add(w, gl_mul(beta, gl_mul(a.k_is[j], x))));
                        dp = gl_mul(dp, gl_add(w, gl_mul(beta, a.cs[(size_t)(a.num_constants + j) * N + p])));
                    }
                    u64 next = k + 1 < nchunks ? a.zs[(size_t)(a.nch + c * a.npp + k) * N + p] : a.zs[(size_t)c * N + p_next];
                    out.emit(gl_sub(gl_mul(prev, np), gl_mul(next, dp)));
                    prev = next;
                }
            }
            for (u32 c = 0; c < a.nch; c++) sum[c] = gl_add(sum[c], out.result((int)c));
            continue;
        }
        p2_gate gate = a.gates[g];
        out.reset(a.nch + a.nch * (a.npp + 1));
        p2_eval_gate(gate, v, a.extra, out);
        u64 f = p2_filter(g, gate.group_start, gate.group_end, v.sel(gate.selector_index), a.nsel > 1);
        for (u32 c = 0; c < a.nch; c++) sum[c] = gl_add(sum[c], gl_mul(f, out.result((int)c)));
    }
    u64 *part = p2_tile + (size_t)ncols * 64;
    for (u32 c = 0; c < a.nch; c++) part[((size_t)wave * P2_MAX_CH + c) * 64 + lane] = sum[c];
    __syncthreads();
    if (wave == 0) {
        u64 i = __brevll((u64)p) >> (64 - a.lde_bits);
        const u32 coset = (u32)i & ((1u << a.rate_bits) - 1);
        for (u32 c = 0; c < a.nch; c++) {
            u64 t = sum[c];
            for (u32 w = 1; w < plan.nwaves; w++) t = gl_add(t, part[((size_t)w * P2_MAX_CH + c) * 64 + lane]);
            a.out[(size_t)c * N + p] = gl_mul(t, a.zh_inv[coset]);
        }
    }
}

// x = g w^bitrev(p) and L_0(x) for every LDE point (circuit constants): the L_0 denominators n (x - 1) are inverted with one
// field inversion per 256 points (Montgomery's trick inside a workgroup's scan would be overkill: one lane = 4 points)
__global__ void __launch_bounds__(256) p2_point_tables_kernel(u64 *xs, u64 *l0, u32 lde_bits, u32 rate_bits, u64 w_lde, u64 n_field,
                                                               p2_quotient_args zhsrc) {
    const size_t N = (size_t)1 << lde_bits;
    size_t p0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (p0 >= N) return;
    u64 x[4], d[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u64 i = __brevll((u64)(p0 + q)) >> (64 - lde_bits);
        x[q] = gl_mul(GL_GENERATOR, gl_pow(w_lde, i));
        d[q] = gl_mul(n_field, gl_sub(x[q], 1));
    }
    u64 p01 = gl_mul(d[0], d[1]), p012 = gl_mul(p01, d[2]), inv = gl_inv(gl_mul(p012, d[3]));
    u64 i3 = gl_mul(inv, p012);
    inv = gl_mul(inv, d[3]);
    u64 i2 = gl_mul(inv, p01);
    inv = gl_mul(inv, d[2]);
    u64 i1 = gl_mul(inv, d[0]), i0 = gl_mul(inv, d[1]);
    u64 di[4] = {i0, i1, i2, i3};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u64 i = __brevll((u64)(p0 + q)) >> (64 - lde_bits);
        xs[p0 + q] = x[q];
        l0[p0 + q] = gl_mul(zhsrc.zh[(u32)i & ((1u << rate_bits) - 1)], di[q]);
    }
}

// ---------------------------------------------------------------------------------------------------- openings
// partial[b][s] = sum over the s-th slice of j of coeffs[b][j] * zpow[j] (* scale[j]);  grid (polynomials, P2_EVAL_SPLIT):
// enough workgroups to fill the chip for 16-234 polynomials, 4 independent loads in flight per lane
#define P2_EVAL_SPLIT 8
__global__ void __launch_bounds__(P2_THREADS)
p2_eval_at_ext_kernel(const u64 *__restrict__ coeffs, u32 n, const gl2 *__restrict__ zpow, const u64 *__restrict__ scale,
                      gl2 *__restrict__ partial) {
    __shared__ gl2 sh[P2_THREADS];
    const u64 *c = coeffs + (size_t)blockIdx.x * n;
    u32 slice = (n + P2_EVAL_SPLIT - 1) / P2_EVAL_SPLIT;
    u32 j0 = blockIdx.y * slice, j1 = j0 + slice < n ? j0 + slice : n;
    gl2 acc = gl2_make(0, 0);
    u32 j = j0 + threadIdx.x;
    for (; j + 3 * P2_THREADS < j1; j += 4 * P2_THREADS) {
        u64 c0 = c[j], c1 = c[j + P2_THREADS], c2 = c[j + 2 * P2_THREADS], c3 = c[j + 3 * P2_THREADS];
        gl2 z0 = zpow[j], z1 = zpow[j + P2_THREADS], z2 = zpow[j + 2 * P2_THREADS], z3 = zpow[j + 3 * P2_THREADS];
        if (scale) {
            c0 = gl_mul(c0, scale[j]);
            c1 = gl_mul(c1, scale[j + P2_THREADS]);
            c2 = gl_mul(c2, scale[j + 2 * P2_THREADS]);
            c3 = gl_mul(c3, scale[j + 3 * P2_THREADS]);
        }
        acc = gl2_add(acc, gl2_add(gl2_add(gl2_scale(z0, c0), gl2_scale(z1, c1)), gl2_add(gl2_scale(z2, c2), gl2_scale(z3, c3))));
    }
    for (; j < j1; j += P2_THREADS) {
        u64 cj = c[j];
        if (scale) cj = gl_mul(cj, scale[j]);
        acc = gl2_add(acc, gl2_scale(zpow[j], cj));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 off = P2_THREADS / 2; off; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] = gl2_add(sh[threadIdx.x], sh[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * P2_EVAL_SPLIT + blockIdx.y] = sh[0];
}
__global__ void p2_eval_finish_kernel(const gl2 *__restrict__ partial, u32 n_polys, gl2 *__restrict__ out) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_polys) return;
    gl2 acc = partial[(size_t)b * P2_EVAL_SPLIT];
    for (u32 s = 1; s < P2_EVAL_SPLIT; s++) acc = gl2_add(acc, partial[(size_t)b * P2_EVAL_SPLIT + s]);
    out[b] = acc;
}

// -------------------------------------------------------------------------------------------------------- FRI
struct p2_fri_combine_args {
    const u64 *mats[4];   // the four committed LDE matrices (bit-reversed), stride N
    u32 widths[4];
    u32 lde_bits, nch;
    u64 w_lde;
    gl2 alpha, zeta, g_zeta, y0, y1, alpha_pow_nch;
    gl2 *out;             // [N] extension elements
    const u32 *apow;      // alpha^i, i < sum(widths), as limbs (p2_ext_pow_limbs_kernel)
};
// fri/oracle.rs `prove_openings` in evaluation form (= fri.go:208-251 on the verifier side):
//   out(x) = alpha^nch * (sum_i alpha^i p_i(x) - y0) / (x - zeta) + (sum_{i<nch} alpha^i z_i(x) - y1) / (x - g zeta)
// The sums are NOT evaluated by Horner's rule (one extension multiplication per polynomial and a dependency chain as long as
// the list): alpha^i comes from a table of 22-bit limbs (p2_ext_pow_limbs_kernel) and every p_i(x) alpha^i is twelve carry-free
// v_mad_u64_u32 into two column accumulators (gl_acc3_mul) -- ~4 k instructions per point for the 357 polynomials of the
// Ed25519 circuit instead of ~30 k, which leaves the kernel to the 8 bytes per polynomial and point it has to read.
__global__ void __launch_bounds__(P2_THREADS) p2_fri_combine_kernel(p2_fri_combine_args a) {
    const size_t N = (size_t)1 << a.lde_bits;
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    u64 i = __brevll((u64)p) >> (64 - a.lde_bits);
    u64 x = gl_mul(GL_GENERATOR, gl_pow(a.w_lde, i));
    gl_acc3 sa = {0, 0, 0}, sb = {0, 0, 0};
    gl_ktab *k6 = (gl_ktab *)a.apow;
    u32 terms = 0;
    for (int m = 0; m < 4; m++) {
        const u64 *mat = a.mats[m] + p;
        const u32 w = a.widths[m];
        u32 j = 0;
        for (; j + 16 <= w; j += 16, k6 += 16 * 12) {   // sixteen loads in flight
            u64 v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = mat[(size_t)(j + q) * N];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                gl_acc3_mul(sa, v[q], k6 + 12 * q);
                gl_acc3_mul(sb, v[q], k6 + 12 * q + 6);
            }
            if ((terms += 16) >= 384) {                  // a column holds 2^10 products = 512 terms; the tails below add < 48
                terms = 0;
                gl_acc3_normalize(sa);
                gl_acc3_normalize(sb);
            }
        }
        for (; j < w; j++, k6 += 12) {
            u64 v = mat[(size_t)j * N];
            gl_acc3_mul(sa, v, k6);
            gl_acc3_mul(sb, v, k6 + 6);
        }
        terms += 16;
    }
    gl2 acc = gl2_make(gl_acc3_reduce(sa), gl_acc3_reduce(sb));
    gl_acc3 ta = {0, 0, 0}, tb = {0, 0, 0};
    for (u32 j = 0; j < a.nch; j++) {
        u64 v = a.mats[2][(size_t)j * N + p];
        gl_acc3_mul(ta, v, (gl_ktab *)a.apow + 12 * (size_t)j);
        gl_acc3_mul(tb, v, (gl_ktab *)a.apow + 12 * (size_t)j + 6);
    }
    gl2 acc1 = gl2_make(gl_acc3_reduce(ta), gl_acc3_reduce(tb));
    gl2 q0 = gl2_mul(gl2_sub(acc, a.y0), gl2_inv(gl2_sub(gl2_make(x, 0), a.zeta)));
    gl2 q1 = gl2_mul(gl2_sub(acc1, a.y1), gl2_inv(gl2_sub(gl2_make(x, 0), a.g_zeta)));
    a.out[p] = gl2_add(gl2_mul(q0, a.alpha_pow_nch), q1);
}

// One FRI reduction in evaluation form.  in: 2^log_m values on shift*<w_M>, bit-reversed; chunk l (2^a consecutive
// values) holds P on the coset x*<w_A>, x = shift * w_M^bitrev(l) (value e at x * w_A^bitrev_a(e)).  With
// P(X) = sum_i X^i P_i(X^A):  c_i = (1/A) sum_j P(x w_A^j) w_A^(-ij) = x^i P_i(x^A)  and the folded value is
// sum_i beta^i P_i(x^A) = sum_i (beta / x)^i c_i   (fri/prover.rs folds the coefficients; fri.go:314-384 interpolates).
__global__ void __launch_bounds__(P2_THREADS)
p2_fri_fold_kernel(const gl2 *__restrict__ in, gl2 *__restrict__ out, u32 log_m, u32 a_bits, u64 shift, u64 w_m, u64 w_a_inv,
                   u64 a_inv, gl2 beta) {
    __shared__ u64 tw[64];
    const u32 A = 1u << a_bits;
    if (threadIdx.x < A) tw[threadIdx.x] = gl_pow(w_a_inv, threadIdx.x);
    __syncthreads();
    u32 chunks = 1u << (log_m - a_bits);
    u32 l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= chunks) return;
    u32 m = log_m == a_bits ? 0 : (u32)(__brevll((u64)l) >> (64 - (log_m - a_bits)));
    u64 x = gl_mul(shift, gl_pow(w_m, m));
    gl2 r = gl2_scale(beta, gl_inv(x));
    const gl2 *v = in + (size_t)l * A;
    gl2 acc = gl2_make(0, 0);
    for (u32 i = A; i-- > 0;) {
        gl2 c = gl2_make(0, 0);
        for (u32 e = 0; e < A; e++) {
            u32 j = __brev(e) >> (32 - a_bits);
            c = gl2_add(c, gl2_scale(v[e], tw[(i * j) & (A - 1)]));
        }
        acc = gl2_add(gl2_mul(acc, r), c);
    }
    out[l] = gl2_scale(acc, a_inv);
}

// final polynomial: coefficients of the last oracle (2^log_m values on shift*<w_M>, bit-reversed), naive O(M^2)
__global__ void __launch_bounds__(P2_THREADS)
p2_fri_final_poly_kernel(const gl2 *__restrict__ in, gl2 *__restrict__ out, u32 log_m, u64 shift_inv, u64 w_m_inv, u64 m_inv) {
    u32 M = 1u << log_m;
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    u64 wk = gl_pow(w_m_inv, k);
    gl2 acc = gl2_make(0, 0);
    u64 t = 1;
    for (u32 i = 0; i < M; i++) {
        u32 pos = log_m ? (u32)(__brevll((u64)i) >> (64 - log_m)) : 0;
        acc = gl2_add(acc, gl2_scale(in[pos], t));
        t = gl_mul(t, wk);
    }
    out[k] = gl2_scale(acc, gl_mul(m_inv, gl_pow(shift_inv, k)));
}

// proof of work (fri/prover.rs `fri_proof_of_work`; verifier check fri.go:75-80): lowest witness w such that the
// challenger, after observing w, answers with a challenge < 2^(64 - pow_bits)
struct p2_pow_args {
    u64 state[12];
    u64 in[8];
    u32 n_in, pow_bits;
    u64 base;
    unsigned long long *found;
};
__global__ void __launch_bounds__(P2_THREADS) p2_pow_kernel(p2_pow_args a) {
    u64 w = a.base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = a.state[i];
#pragma unroll
    for (int i = 0; i < 8; i++)
        if ((u32)i < a.n_in) s[i] = a.in[i];
#pragma unroll
    for (int i = 0; i < 8; i++)
        if ((u32)i == a.n_in) s[i] = w;
    poseidon_gl_permute(s);
    if ((s[7] >> (64 - a.pow_bits)) == 0) atomicMin(a.found, (unsigned long long)w);
}

// query openings: strided gathers described by the host
struct p2_gather {
    const u64 *src;
    u64 stride;
    u32 count, dst;
};
__global__ void p2_gather_kernel(const p2_gather *__restrict__ d, u32 nd, u64 *__restrict__ out) {
    u32 g = blockIdx.x;
    if (g >= nd) return;
    p2_gather e = d[g];
    for (u32 i = threadIdx.x; i < e.count; i += blockDim.x) out[e.dst + i] = e.src[(size_t)i * e.stride];
}

// ----------------------------------------------------------------------------------------------------- host side
static u64 h_mul(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % GL_P); }
static u64 h_add(u64 a, u64 b) { return (u64)(((unsigned __int128)a + b) % GL_P); }
static u64 h_sub(u64 a, u64 b) { return a >= b ? a - b : a + GL_P - b; }
static u64 h_pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = h_mul(r, a);
        a = h_mul(a, a);
        e >>= 1;
    }
    return r;
}
static u64 h_inv(u64 a) { return h_pow(a, GL_P - 2); }
static u64 h_root(u32 log_n) { return h_pow(GL_POWER_OF_TWO_GENERATOR, 1ULL << (32 - log_n)); }
static gl2 h2_mul(gl2 x, gl2 y) {
    return gl2_make(h_add(h_mul(x.a, y.a), h_mul(7, h_mul(x.b, y.b))), h_add(h_mul(x.a, y.b), h_mul(x.b, y.a)));
}
static gl2 h2_add(gl2 x, gl2 y) { return gl2_make(h_add(x.a, y.a), h_add(x.b, y.b)); }
static gl2 h2_pow(gl2 x, u64 e) {
    gl2 r = gl2_make(1, 0);
    while (e) {
        if (e & 1) r = h2_mul(r, x);
        x = h2_mul(x, x);
        e >>= 1;
    }
    return r;
}

struct p2_batch {
    u32 width = 0;
    u64 *coeffs = nullptr, *lde = nullptr, *tree = nullptr;
};

struct zklc_plonky2_circuit {
    zklc_ctx *ctx = nullptr;
    int device = 0;   // copied from the context: destroying the circuit must not touch a context that is already gone
    zklc_plonky2_params P;
    u32 n = 0, N = 0, lde_bits = 0, nchunks = 0;
    std::vector<p2_gate> gates;
    p2_gate *d_gates = nullptr;
    u64 *d_extra = nullptr, *d_kis = nullptr, *d_subgroup = nullptr, *d_sigma_vals = nullptr;
    p2_batch cs, wires, zs, quot;
    u64 *d_wire_vals = nullptr;       // witness values (routed columns are needed after the iNTT)
    u64 *d_rp = nullptr, *d_excl = nullptr, *d_totals = nullptr, *d_grand = nullptr;
    u64 *d_qv = nullptr;              // quotient values [nch][N]
    u32 *d_apow = nullptr;            // powers of the alphas for the quotient kernels (6 limb words per power)
    u32 *d_fri_apow = nullptr;        // powers of the FRI batching challenge (12 limb words per power)
    u64 *d_xs = nullptr, *d_l0 = nullptr;   // per LDE point: x and L_0(x) (p2_point_tables_kernel)
    p2_fused_plan plan = {};          // job lists of the fused quotient kernel; plan.nwaves == 0: per-gate launches
    size_t fused_lds = 0;
    std::vector<double> gate_ms;      // calibration: per-gate kernel time (the last entry is the base kernel)
    gl2 *d_zpow = nullptr, *d_open = nullptr, *d_open_partial = nullptr;
    gl2 *d_fri[9] = {};               // FRI oracles (extension values), [0] has N elements
    u64 *d_fri_tree[8] = {};
    gl2 *d_final = nullptr;
    unsigned long long *d_found = nullptr;
    p2_gather *d_gather = nullptr;
    u64 *d_gather_out = nullptr;
    size_t gather_cap = 0, gather_out_words = 0;
    std::vector<uint8_t> cap_bytes;
    uint8_t digest[32];
    std::vector<u64> last_challenges;
    double timings[8] = {};
    u64 tree_words = 0;
    std::vector<void *> allocs;
    // page-locked staging for every host <-> device transfer of a proof (p2_pin): blocks, bytes used of the last one
    std::vector<std::pair<uint8_t *, size_t>> pin_blocks;
    size_t pin_used = 0;
};

// Pinned staging.  hipMemcpyAsync to or from PAGEABLE host memory is not an enqueue: the runtime stages the bytes itself and the
// call returns only when the stream has reached the copy -- waiting ACTIVELY.  With the transcript on the host a proof has ~10 such
// points, so a proving thread sat at 1.00 host cores for the whole proof (profiles/r05d_host_cpu_probe.txt) although every explicit
// wait already went through an event created with hipEventBlockingSync (zklc_stream_wait).  Round 6: every transfer of a proof goes
// through this per-circuit page-locked arena, hipMemcpyAsync returns at once, and the thread sleeps in zklc_stream_wait -- blocking
// waits by construction, no device-wide flag (the hipSetDeviceFlags option of round 5 is gone).  The arena is reset at the start of
// a proof (a circuit proves one witness at a time) and grows by whole blocks, so pointers handed out stay valid until the next proof.
static bool p2_pin_enabled() {
    static const bool on = !(getenv("ZKLC_PINNED_STAGING") && getenv("ZKLC_PINNED_STAGING")[0] == '0');      // A/B switch
    return on;
}
static void p2_pin_free(void *q) {
    if (p2_pin_enabled()) (void)hipHostFree(q);
    else free(q);
}
static void *p2_pin(zklc_plonky2_circuit *c, size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    if (c->pin_blocks.empty() || c->pin_used + bytes > c->pin_blocks.back().second) {
        size_t cap = bytes > (1u << 20) ? bytes : (1u << 20);
        void *q = nullptr;
        if (!p2_pin_enabled()) {
            q = malloc(cap);                                   // A/B: pageable staging, as rounds 1-5
            if (!q) return nullptr;
        } else if (hipHostMalloc(&q, cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        c->pin_blocks.push_back({(uint8_t *)q, cap});
        c->pin_used = 0;
    }
    void *r = c->pin_blocks.back().first + c->pin_used;
    c->pin_used += bytes;
    return r;
}
// start of a proof: one block of the size the last proof needed (no transfer of this circuit is in flight here)
static void p2_pin_reset(zklc_plonky2_circuit *c) {
    if (c->pin_blocks.size() > 1) {
        size_t total = 0;
        for (auto &b : c->pin_blocks) {
            total += b.second;
            p2_pin_free(b.first);
        }
        c->pin_blocks.clear();
        void *q = nullptr;
        if (!p2_pin_enabled()) {
            if ((q = malloc(total))) c->pin_blocks.push_back({(uint8_t *)q, total});
        } else if (hipHostMalloc(&q, total, hipHostMallocDefault) == hipSuccess) c->pin_blocks.push_back({(uint8_t *)q, total});
        else (void)hipGetLastError();
    }
    c->pin_used = 0;
}
#define P2_PIN(c, ptr, type, count)                                          \
    type *ptr = (type *)p2_pin((c), sizeof(type) * (size_t)(count));          \
    if (!ptr) {                                                               \
        (c)->ctx->last_err = "hipHostMalloc: pinned staging";                 \
        return ZKLC_ERR_OOM;                                                  \
    }

static int32_t p2_alloc(zklc_plonky2_circuit *c, void **p, size_t bytes) {
    zklc_ctx *ctx = c->ctx;
    ZKLC_HIP(ctx, hipMalloc(p, bytes ? bytes : 8));
    c->allocs.push_back(*p);
    return ZKLC_OK;
}
#define P2_ALLOC(c, ptr, bytes)                                      \
    do {                                                             \
        int32_t rc__ = p2_alloc((c), (void **)&(ptr), (bytes));      \
        if (rc__) return rc__;                                       \
    } while (0)
#define P2_RC(call)               \
    do {                          \
        int32_t rc__ = (call);    \
        if (rc__) return rc__;    \
    } while (0)

static inline u64 *p2_tree_level(u64 *tree, u32 log_leaves, u32 level) {
    u64 off = 0;
    for (u32 j = 0; j < level; j++) off += 4ULL << (log_leaves - j);
    return tree + off;
}

static int32_t p2_merkle(zklc_plonky2_circuit *c, hipStream_t st, const u64 *mat, u64 stride, u64 leaf_stride, u32 log_leaves,
                         u32 width, u64 *tree) {
    u32 cap_h = c->P.cap_height < log_leaves ? c->P.cap_height : log_leaves;
    if (c->P.hasher == ZKLC_HASHER_POSEIDON_GL)
        return zklc_gl_merkle_commit_strided(c->ctx, st, mat, stride, leaf_stride, log_leaves, width, cap_h, tree);
    return zklc_bn254_merkle_commit_strided(c->ctx, st, mat, stride, leaf_stride, log_leaves, width, cap_h, tree);
}

// coefficients (natural order, [width][n]) -> LDE (bit-reversed) -> Merkle tree; cap copied to the host
static int32_t p2_commit_coeffs(zklc_plonky2_circuit *c, hipStream_t st, p2_batch &b, std::vector<uint8_t> &cap) {
    zklc_ctx *ctx = c->ctx;
    P2_RC(zklc_gl_lde_dev(ctx, st, b.coeffs, c->P.degree_bits, c->P.rate_bits, b.width, GL_GENERATOR, b.lde, ZKLC_NTT_OUT_BITREV));
    P2_RC(p2_merkle(c, st, b.lde, c->N, 1, c->lde_bits, b.width, b.tree));
    u32 cap_h = c->P.cap_height < c->lde_bits ? c->P.cap_height : c->lde_bits;
    cap.resize((size_t)32 << cap_h);
    P2_PIN(c, h_cap, uint8_t, cap.size());
    ZKLC_HIP(ctx, zklc_readback_async(h_cap, p2_tree_level(b.tree, c->lde_bits, c->lde_bits - cap_h), cap.size(), st));
    ZKLC_HIP(ctx, zklc_stream_wait(st));
    memcpy(cap.data(), h_cap, cap.size());
    return ZKLC_OK;
}
// values on the subgroup (natural order, in b.coeffs) -> coefficients in place, then commit
static int32_t p2_commit_values(zklc_plonky2_circuit *c, hipStream_t st, p2_batch &b, std::vector<uint8_t> &cap) {
    P2_RC(zklc_gl_ntt_dev(c->ctx, st, b.coeffs, c->P.degree_bits, b.width, ZKLC_NTT_INVERSE, 0));
    return p2_commit_coeffs(c, st, b, cap);
}

// C::Hasher::hash_no_pad of a short Goldilocks vector -> 32 bytes
static int32_t p2_hash_no_pad(zklc_plonky2_circuit *c, hipStream_t st, const std::vector<u64> &v, uint8_t *out32) {
    zklc_ctx *ctx = c->ctx;
    if (c->P.hasher == ZKLC_HASHER_POSEIDON_GL) {
        u64 h[4];
        zklc_host_poseidon_hash_no_pad(v.data(), v.size(), h);
        memcpy(out32, h, 32);
        return ZKLC_OK;
    }
    // Poseidon-BN128: one leaf of width len > 3 through the leaf kernel (hash_or_noop == hash_no_pad for len > 3)
    if (v.size() <= 3) return ZKLC_ERR_INVALID_ARG;
    void *d;
    P2_RC(zklc_stage(ctx, 6, v.size() * 8 + 64, &d));
    u64 *dv = (u64 *)d;
    P2_PIN(c, h_v, u64, v.size() + 4);
    memcpy(h_v + 4, v.data(), v.size() * 8);
    ZKLC_HIP(ctx, hipMemcpyAsync(dv + 8, h_v + 4, v.size() * 8, hipMemcpyHostToDevice, st));
    P2_RC(zklc_bn254_merkle_commit_strided(ctx, st, dv + 8, 1, 0, 0, (u32)v.size(), 0, dv));
    ZKLC_HIP(ctx, zklc_readback_async(h_v, dv, 32, st));
    ZKLC_HIP(ctx, zklc_stream_wait(st));
    memcpy(out32, h_v, 32);
    return ZKLC_OK;
}

static void p2_quotient_static_args(const zklc_plonky2_circuit *c, p2_quotient_args &a) {
    const zklc_plonky2_params &P = c->P;
    a.cs = c->cs.lde;
    a.wires = c->wires.lde;
    a.zs = c->zs.lde;
    a.gates = c->d_gates;
    a.extra = c->d_extra;
    a.k_is = c->d_kis;
    a.lde_bits = c->lde_bits;
    a.degree_bits = P.degree_bits;
    a.rate_bits = P.rate_bits;
    a.num_constants = P.num_constants;
    a.nsel = P.num_selectors;
    a.routed = P.num_routed_wires;
    a.nch = P.num_challenges;
    a.npp = P.num_partial_products;
    a.qdf = P.quotient_degree_factor;
    a.num_gates = P.num_gates;
    a.num_wires = P.num_wires;
    a.w_lde = h_root(c->lde_bits);
    u64 gn = h_pow(GL_GENERATOR, c->n), w_r = h_root(P.rate_bits);
    for (u32 k = 0; k < (1u << P.rate_bits); k++) {
        a.zh[k] = h_sub(h_mul(gn, h_pow(w_r, k)), 1);
        a.zh_inv[k] = h_inv(a.zh[k]);
    }
    a.n_field = c->n;
    a.xs = c->d_xs;
    a.l0 = c->d_l0;
    a.out = c->d_qv;
}

// Measures every gate's (and the base terms') kernel once on the circuit's own buffers and splits the jobs over the waves of
// the fused kernel: longest job first onto the least loaded wave.  ZKLC_P2_QUOTIENT=pergate keeps the per-gate launches,
// ZKLC_P2_FQ_WAVES overrides the number of waves per workgroup (default: 8 when only one workgroup's tile fits a CU's LDS).
static int32_t p2_plan_quotient(zklc_plonky2_circuit *c, hipStream_t st) {
    zklc_ctx *ctx = c->ctx;
    const zklc_plonky2_params &P = c->P;
    const u32 N = c->N;
    c->plan.nwaves = 0;
    // measured on the Ed25519 circuit (profiles/r02_quotient_ab.txt): per-gate launches 15.4 ms of kernels, fused 30 ms with 8
    // waves per workgroup, 51 ms with 4 -- the 120 KB tile leaves 1-2 waves per SIMD and the evaluators (long dependent chains
    // of multiply-reduce) need 6-8 to reach the VALU issue rate.  The fused form therefore stays opt-in.
    const char *mode = getenv("ZKLC_P2_QUOTIENT");
    if (!mode || strcmp(mode, "fused")) return ZKLC_OK;
    const size_t tile = (size_t)(P.num_constants + P.num_wires) * 64 * 8;
    if (N < 64 || P.num_gates + 1 > P2_FQ_MAX_WAVES * P2_FQ_MAX_JOBS) return ZKLC_OK;
    u32 nw = tile * 2 + 2 * 4 * P2_MAX_CH * 512 <= 160 * 1024 ? 4 : 8;
    if (const char *e = getenv("ZKLC_P2_FQ_WAVES")) nw = (u32)atoi(e);
    if (nw < 1 || nw > P2_FQ_MAX_WAVES) return ZKLC_ERR_INVALID_ARG;
    c->fused_lds = tile + (size_t)nw * P2_MAX_CH * 64 * 8;
    if (c->fused_lds > 160 * 1024) return ZKLC_OK;      // the tile does not fit: per-gate launches
    ZKLC_HIP(ctx, hipFuncSetAttribute((const void *)p2_quotient_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)c->fused_lds));
    p2_quotient_args a = {};
    p2_quotient_static_args(c, a);
    for (u32 k = 0; k < P2_MAX_CH; k++) a.apow[k] = c->d_apow;
    hipEvent_t e0, e1;
    ZKLC_HIP(ctx, hipEventCreate(&e0));
    ZKLC_HIP(ctx, hipEventCreate(&e1));
    c->gate_ms.assign(P.num_gates + 1, 0.0);
    for (u32 g = 0; g <= P.num_gates; g++) {
        p2_gate_kernel_fn fn = g < P.num_gates ? p2_gate_kernel_of(c->gates[g].type) : nullptr;
        if (g < P.num_gates && !fn) continue;
        for (int rep = 0; rep < 2; rep++) {
            ZKLC_HIP(ctx, hipEventRecord(e0, st));
            if (fn) {
                p2_gate_list one = {};
                one.n = 1;
                one.idx[0] = g;
                hipLaunchKernelGGL(fn, dim3((N + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st, a, one);
            } else
                hipLaunchKernelGGL(p2_quotient_base_kernel, dim3((N + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st, a);
            ZKLC_HIP(ctx, hipEventRecord(e1, st));
            ZKLC_HIP(ctx, hipEventSynchronize(e1));
            float ms = 0;
            ZKLC_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
            if (rep == 0 || ms < c->gate_ms[g]) c->gate_ms[g] = ms;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    std::vector<u32> order;
    for (u32 g = 0; g <= P.num_gates; g++)
        if (g == P.num_gates || p2_gate_kernel_of(c->gates[g].type)) order.push_back(g);
    std::sort(order.begin(), order.end(), [&](u32 x, u32 y) { return c->gate_ms[x] > c->gate_ms[y]; });
    double load[P2_FQ_MAX_WAVES] = {};
    p2_fused_plan plan = {};
    for (u32 g : order) {
        u32 best = 0;
        for (u32 w = 1; w < nw; w++)
            if (load[w] < load[best]) best = w;
        if (plan.njobs[best] >= P2_FQ_MAX_JOBS) return ZKLC_OK;
        plan.jobs[best][plan.njobs[best]++] = g == P.num_gates ? (uint16_t)P2_JOB_BASE : (uint16_t)g;
        load[best] += c->gate_ms[g];
    }
    plan.nwaves = nw;
    c->plan = plan;
    if (getenv("ZKLC_P2_DEBUG")) {
        fprintf(stderr, "[zklc] fused quotient: %u waves, LDS %zu B; per-gate calibration (ms):", nw, c->fused_lds);
        for (u32 g = 0; g <= P.num_gates; g++) fprintf(stderr, " %u:%.3f", g < P.num_gates ? c->gates[g].type : 99u, c->gate_ms[g]);
        fprintf(stderr, "\n[zklc] wave loads (ms):");
        for (u32 w = 0; w < nw; w++) fprintf(stderr, " %.3f", load[w]);
        fprintf(stderr, "\n");
    }
    return ZKLC_OK;
}

extern "C" void zklc_plonky2_circuit_destroy(zklc_plonky2_circuit *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (void *p : c->allocs) (void)hipFree(p);
    for (auto &b : c->pin_blocks) p2_pin_free(b.first);
    delete c;
}

static int32_t p2_create(zklc_ctx *ctx, const zklc_plonky2_params *params, const zklc_plonky2_gate *gates, const uint64_t *gate_extra,
                         uint32_t gate_extra_words, const uint64_t *k_is, const uint64_t *constants, const uint64_t *sigmas,
                         zklc_plonky2_circuit *c) {
    const zklc_plonky2_params &P = *params;
    c->ctx = ctx;
    c->device = ctx->device;
    c->P = P;
    c->n = 1u << P.degree_bits;
    c->lde_bits = P.degree_bits + P.rate_bits;
    c->N = 1u << c->lde_bits;
    c->nchunks = P.num_partial_products + 1;
    const u32 n = c->n, N = c->N;
    hipStream_t st = ctx->stream;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    c->gates.resize(P.num_gates);
    for (u32 i = 0; i < P.num_gates; i++) {
        p2_gate g;
        g.type = gates[i].type;
        for (int k = 0; k < 4; k++) g.p[k] = gates[i].p[k];
        g.selector_index = gates[i].selector_index;
        g.group_start = gates[i].group_start;
        g.group_end = gates[i].group_end;
        g.extra_off = gates[i].extra_off;
        if (g.type >= P2_NUM_GATE_TYPES || g.selector_index >= P.num_selectors) return ZKLC_ERR_INVALID_ARG;
        if (g.type == P2_RANDOM_ACCESS && g.p[0] > 6) return ZKLC_ERR_INVALID_ARG;
        if (g.type == P2_COSET_INTERPOLATION && (g.p[0] > 5 || g.p[1] < 2 || g.extra_off + (2u << g.p[0]) > gate_extra_words))
            return ZKLC_ERR_INVALID_ARG;
        c->gates[i] = g;
    }
    P2_ALLOC(c, c->d_gates, sizeof(p2_gate) * P.num_gates);
    ZKLC_HIP(ctx, hipMemcpyAsync(c->d_gates, c->gates.data(), sizeof(p2_gate) * P.num_gates, hipMemcpyHostToDevice, st));
    P2_ALLOC(c, c->d_extra, (size_t)gate_extra_words * 8);
    if (gate_extra_words) ZKLC_HIP(ctx, hipMemcpyAsync(c->d_extra, gate_extra, (size_t)gate_extra_words * 8, hipMemcpyHostToDevice, st));
    P2_ALLOC(c, c->d_kis, (size_t)P.num_routed_wires * 8);
    ZKLC_HIP(ctx, hipMemcpyAsync(c->d_kis, k_is, (size_t)P.num_routed_wires * 8, hipMemcpyHostToDevice, st));
    const u32 nch_ = P.num_challenges;
    P2_ALLOC(c, c->d_subgroup, (size_t)n * 8);
    hipLaunchKernelGGL(p2_pow_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->d_subgroup, h_root(P.degree_bits), (u64)n);
    ZKLC_HIP(ctx, hipGetLastError());

    c->tree_words = zklc_gl_merkle_tree_words(c->lde_bits, P.cap_height < c->lde_bits ? P.cap_height : c->lde_bits);
    auto alloc_batch = [&](p2_batch &b, u32 width) -> int32_t {
        b.width = width;
        P2_ALLOC(c, b.coeffs, (size_t)width * n * 8);
        P2_ALLOC(c, b.lde, (size_t)width * N * 8);
        P2_ALLOC(c, b.tree, c->tree_words * 8);
        return ZKLC_OK;
    };
    const u32 nch = P.num_challenges;
    P2_RC(alloc_batch(c->cs, P.num_constants + P.num_routed_wires));
    P2_RC(alloc_batch(c->wires, P.num_wires));
    P2_RC(alloc_batch(c->zs, nch * (1 + P.num_partial_products)));
    // quotient coefficients are produced in place in d_qv ([nch][N] == [nch * qdf][n])
    c->quot.width = nch * P.quotient_degree_factor;
    P2_ALLOC(c, c->d_qv, (size_t)nch * N * 8);
    c->quot.coeffs = c->d_qv;
    P2_ALLOC(c, c->quot.lde, (size_t)c->quot.width * N * 8);
    P2_ALLOC(c, c->quot.tree, c->tree_words * 8);
    P2_ALLOC(c, c->d_wire_vals, (size_t)P.num_wires * n * 8);
    P2_ALLOC(c, c->d_sigma_vals, (size_t)P.num_routed_wires * n * 8);
    P2_ALLOC(c, c->d_rp, (size_t)c->nchunks * n * 8);
    P2_ALLOC(c, c->d_excl, (size_t)n * 8);
    P2_ALLOC(c, c->d_totals, (size_t)((n + P2_SCAN_BLOCK - 1) / P2_SCAN_BLOCK) * 8);
    P2_ALLOC(c, c->d_grand, 8);
    P2_ALLOC(c, c->d_apow, (size_t)nch_ * (nch_ + nch_ * (P.num_partial_products + 1) + P.num_gate_constraints + 1) * 24);
    P2_ALLOC(c, c->d_zpow, (size_t)n * sizeof(gl2));
    P2_ALLOC(c, c->d_fri_apow, (size_t)(c->cs.width + c->wires.width + c->zs.width + c->quot.width + 1) * 12 * sizeof(u32));
    P2_ALLOC(c, c->d_xs, (size_t)N * 8);
    P2_ALLOC(c, c->d_l0, (size_t)N * 8);
    u32 total_polys = c->cs.width + c->wires.width + c->zs.width + c->quot.width + nch;
    P2_ALLOC(c, c->d_open, (size_t)total_polys * sizeof(gl2));
    P2_ALLOC(c, c->d_open_partial, (size_t)total_polys * P2_EVAL_SPLIT * sizeof(gl2));
    // FRI oracles
    u32 bits = c->lde_bits;
    P2_ALLOC(c, c->d_fri[0], (size_t)N * sizeof(gl2));
    for (u32 r = 0; r < P.num_arities; r++) {
        u32 leaves_bits = bits - P.arity_bits[r];
        u32 cap_h = P.cap_height < leaves_bits ? P.cap_height : leaves_bits;
        P2_ALLOC(c, c->d_fri_tree[r], zklc_gl_merkle_tree_words(leaves_bits, cap_h) * 8);
        bits = leaves_bits;
        P2_ALLOC(c, c->d_fri[r + 1], ((size_t)1 << bits) * sizeof(gl2));
    }
    P2_ALLOC(c, c->d_final, ((size_t)1 << bits) * sizeof(gl2));
    P2_ALLOC(c, c->d_found, 8);

    // preprocessed polynomials: constants then sigmas (values) -> coefficients -> LDE -> Merkle
    size_t cbytes = (size_t)P.num_constants * n * 8, sbytes = (size_t)P.num_routed_wires * n * 8;
    ZKLC_HIP(ctx, hipMemcpyAsync(c->cs.coeffs, constants, cbytes, hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, hipMemcpyAsync(c->cs.coeffs + (size_t)P.num_constants * n, sigmas, sbytes, hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, hipMemcpyAsync(c->d_sigma_vals, sigmas, sbytes, hipMemcpyHostToDevice, st));
    P2_RC(p2_commit_values(c, st, c->cs, c->cap_bytes));
    {
        p2_quotient_args a = {};
        p2_quotient_static_args(c, a);
        hipLaunchKernelGGL(p2_point_tables_kernel, dim3((N / 4 + 255) / 256 ? (N / 4 + 255) / 256 : 1), dim3(256), 0, st, c->d_xs, c->d_l0,
                           c->lde_bits, P.rate_bits, a.w_lde, a.n_field, a);
        ZKLC_HIP(ctx, hipGetLastError());
        P2_RC(p2_plan_quotient(c, st));
    }

    // circuit digest = hash_no_pad(cap as field elements || hash_pad([]) || degree_bits)  (plonky2 circuit_builder.rs `build`)
    auto hash_to_vec = [&](const uint8_t *h, std::vector<u64> &out) {
        if (P.hasher == ZKLC_HASHER_POSEIDON_GL) {
            u64 v[4];
            memcpy(v, h, 32);
            out.insert(out.end(), v, v + 4);
        } else {
            for (int off = 0; off < 32; off += 7) {
                u64 v = 0;
                memcpy(&v, h + off, 32 - off < 7 ? 32 - off : 7);
                out.push_back(v);
            }
        }
    };
    std::vector<u64> parts;
    for (size_t i = 0; i < c->cap_bytes.size(); i += 32) hash_to_vec(c->cap_bytes.data() + i, parts);
    std::vector<u64> pad;
    pad.push_back(1);
    u32 pad_to = P.hasher == ZKLC_HASHER_POSEIDON_GL ? 8 : 9;
    while ((pad.size() + 1) % pad_to) pad.push_back(0);
    pad.push_back(1);
    uint8_t dsep[32];
    P2_RC(p2_hash_no_pad(c, st, pad, dsep));
    hash_to_vec(dsep, parts);
    parts.push_back(P.degree_bits);
    P2_RC(p2_hash_no_pad(c, st, parts, c->digest));
    return ZKLC_OK;
}

extern "C" int32_t zklc_plonky2_circuit_create(zklc_ctx *ctx, const zklc_plonky2_params *params, const zklc_plonky2_gate *gates,
                                               const uint64_t *gate_extra, uint32_t gate_extra_words, const uint64_t *k_is,
                                               const uint64_t *constants, const uint64_t *sigmas, zklc_plonky2_circuit **out) {
    if (!ctx || !params || !gates || !k_is || !constants || !sigmas || !out) return ZKLC_ERR_INVALID_ARG;
    const zklc_plonky2_params &P = *params;
    if (P.num_challenges == 0 || P.num_challenges > P2_MAX_CH || P.degree_bits == 0 || P.degree_bits + P.rate_bits > ZKLC_GL_MAX_LOG ||
        P.rate_bits > 4 || P.quotient_degree_factor != (1u << P.rate_bits) || P.num_arities > 8 || P.hasher > 1 ||
        P.num_routed_wires > P.num_wires || P.num_selectors == 0 || P.num_selectors > P.num_constants || P.num_gates == 0 ||
        P.num_partial_products + 1 > 16 || (P.num_partial_products + 1) * P.quotient_degree_factor < P.num_routed_wires ||
        P.proof_of_work_bits == 0 || P.proof_of_work_bits > 40)
        return ZKLC_ERR_INVALID_ARG;
    u32 bits = P.degree_bits + P.rate_bits;
    for (u32 r = 0; r < P.num_arities; r++) {
        if (P.arity_bits[r] == 0 || P.arity_bits[r] > 5 || P.arity_bits[r] > bits) return ZKLC_ERR_INVALID_ARG;
        bits -= P.arity_bits[r];
    }
    if (bits < P.rate_bits) return ZKLC_ERR_INVALID_ARG;
    zklc_plonky2_circuit *c = new (std::nothrow) zklc_plonky2_circuit();
    if (!c) return ZKLC_ERR_OOM;
    int32_t rc = p2_create(ctx, params, gates, gate_extra, gate_extra_words, k_is, constants, sigmas, c);
    if (rc) {
        zklc_plonky2_circuit_destroy(c);
        return rc;
    }
    *out = c;
    return ZKLC_OK;
}

extern "C" int32_t zklc_plonky2_verifier_data(zklc_plonky2_circuit *c, uint8_t *cap_out, uint8_t *digest_out) {
    if (!c || !cap_out || !digest_out) return ZKLC_ERR_INVALID_ARG;
    memcpy(cap_out, c->cap_bytes.data(), c->cap_bytes.size());
    memcpy(digest_out, c->digest, 32);
    return ZKLC_OK;
}

// ---- proof layout (plonky2 util/serialization.rs `write_proof_with_public_inputs`)
struct p2_layout {
    u32 cap_n, depth0;
    size_t per_round_words;    // u64 words gathered per query round (leaves, siblings, step evals, step siblings)
    size_t bytes;
};
static p2_layout p2_proof_layout(const zklc_plonky2_circuit *c) {
    const zklc_plonky2_params &P = c->P;
    p2_layout L;
    u32 cap_h = P.cap_height < c->lde_bits ? P.cap_height : c->lde_bits;
    L.cap_n = 1u << cap_h;
    L.depth0 = c->lde_bits - cap_h;
    u32 widths[4] = {c->cs.width, c->wires.width, c->zs.width, c->quot.width};
    size_t openings = c->cs.width + c->wires.width + c->zs.width + c->quot.width + P.num_challenges;
    size_t bytes = 3 * (size_t)L.cap_n * 32 + openings * 16;
    size_t per_bytes = 0, per_words = 0;
    for (int k = 0; k < 4; k++) {
        per_bytes += 8 * (size_t)widths[k] + 1 + 32 * (size_t)L.depth0;
        per_words += widths[k] + 4 * (size_t)L.depth0;
    }
    u32 bits = c->lde_bits;
    for (u32 r = 0; r < P.num_arities; r++) {
        bits -= P.arity_bits[r];
        u32 ch = P.cap_height < bits ? P.cap_height : bits;
        bytes += (size_t)32 << ch;
        per_bytes += 16 * ((size_t)1 << P.arity_bits[r]) + 1 + 32 * (size_t)(bits - ch);
        per_words += 2 * ((size_t)1 << P.arity_bits[r]) + 4 * (size_t)(bits - ch);
    }
    bytes += per_bytes * P.num_query_rounds;
    bytes += 16 * ((size_t)1 << (bits - P.rate_bits)) + 8 + 8 + 8 * (size_t)P.num_public_inputs;
    L.per_round_words = per_words;
    L.bytes = bytes;
    return L;
}

extern "C" uint64_t zklc_plonky2_proof_bytes(zklc_plonky2_circuit *c) { return c ? p2_proof_layout(c).bytes : 0; }

extern "C" uint32_t zklc_plonky2_last_challenges(zklc_plonky2_circuit *c, uint64_t *out, uint32_t cap) {
    if (!c || !out) return 0;
    u32 k = (u32)c->last_challenges.size() < cap ? (u32)c->last_challenges.size() : cap;
    memcpy(out, c->last_challenges.data(), (size_t)k * 8);
    return k;
}
extern "C" uint32_t zklc_plonky2_last_timings(zklc_plonky2_circuit *c, double *out_ms, uint32_t cap) {
    if (!c || !out_ms) return 0;
    u32 k = cap < 8 ? cap : 8;
    memcpy(out_ms, c->timings, (size_t)k * sizeof(double));
    return k;
}

static double p2_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void p2_observe_cap(zklc_challenger &ch, const std::vector<uint8_t> &cap, int hasher) {
    for (size_t i = 0; i < cap.size(); i += 32) ch.observe_hash(cap.data() + i, hasher);
}

extern "C" int32_t zklc_plonky2_prove_dev(zklc_ctx *ctx, void *stream, zklc_plonky2_circuit *c, const uint64_t *d_wires,
                                          const uint64_t *public_inputs, uint8_t *proof_out, uint64_t proof_cap, uint64_t *proof_len) {
    if (!ctx || !c || c->ctx != ctx || !d_wires || !proof_out || !proof_len) return ZKLC_ERR_INVALID_ARG;
    const zklc_plonky2_params &P = c->P;
    if (P.num_public_inputs && !public_inputs) return ZKLC_ERR_INVALID_ARG;
    const p2_layout L = p2_proof_layout(c);
    if (proof_cap < L.bytes) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = zklc_pick_stream(ctx, stream);
    const u32 n = c->n, N = c->N, nch = P.num_challenges, npp = P.num_partial_products;
    const int hasher = (int)P.hasher;
    double t0 = p2_now_ms(), t_prev = t0;
    auto lap = [&](int slot) {
        double t = p2_now_ms();
        c->timings[slot] = t - t_prev;
        t_prev = t;
    };
    c->last_challenges.clear();
    p2_pin_reset(c);

    // ---- transcript start
    u64 pih[4];
    zklc_host_poseidon_hash_no_pad(public_inputs, P.num_public_inputs, pih);
    zklc_challenger ch;
    ch.observe_hash(c->digest, hasher);
    ch.observe_many(pih, 4);

    // ---- wires commitment
    std::vector<uint8_t> wires_cap, zs_cap, quot_cap;
    // values -> coefficients out of place (the wire values stay where the caller put them: Z and the partial products read them)
    P2_RC(zklc_gl_intt_copy_dev(ctx, st, d_wires, c->wires.coeffs, P.degree_bits, P.num_wires));
    P2_RC(p2_commit_coeffs(c, st, c->wires, wires_cap));
    p2_observe_cap(ch, wires_cap, hasher);
    p2_challenges chal = {};
    for (u32 k = 0; k < nch; k++) chal.beta[k] = ch.challenge();
    for (u32 k = 0; k < nch; k++) chal.gamma[k] = ch.challenge();
    lap(0);

    // ---- Z and partial products (values written into zs.coeffs, then interpolated in place)
    u32 nblocks = (n + P2_SCAN_BLOCK - 1) / P2_SCAN_BLOCK;
    P2_PIN(c, grand, u64, P2_MAX_CH);
    for (u32 k = 0; k < P2_MAX_CH; k++) grand[k] = 1;
    for (u32 k = 0; k < nch; k++) {
        hipLaunchKernelGGL(p2_chunk_products_kernel, dim3((n + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st, d_wires,
                           (const u64 *)c->d_sigma_vals, (const u64 *)c->d_subgroup, (const u64 *)c->d_kis, n, P.num_routed_wires,
                           P.quotient_degree_factor, c->nchunks, chal.beta[k], chal.gamma[k], c->d_rp);
        hipLaunchKernelGGL(p2_scan_local_kernel, dim3(nblocks), dim3(P2_THREADS), 0, st, (const u64 *)(c->d_rp + (size_t)npp * n),
                           c->d_excl, c->d_totals, n);
        hipLaunchKernelGGL(p2_scan_totals_kernel, dim3(1), dim3(P2_THREADS), 0, st, c->d_totals, nblocks, c->d_grand);
        hipLaunchKernelGGL(p2_z_apply_kernel, dim3((n + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st, (const u64 *)c->d_excl,
                           (const u64 *)c->d_totals, (const u64 *)c->d_rp, n, npp, c->zs.coeffs + (size_t)k * n,
                           c->zs.coeffs + (size_t)(nch + k * npp) * n);
        ZKLC_HIP(ctx, hipGetLastError());
        ZKLC_HIP(ctx, zklc_readback_async(&grand[k], c->d_grand, 8, st));
    }
    P2_RC(p2_commit_values(c, st, c->zs, zs_cap));
    for (u32 k = 0; k < nch; k++)
        if (grand[k] != 1) {
            ctx->last_err = "plonky2: the witness violates the copy constraints (permutation product != 1)";
            return ZKLC_ERR_INVALID_ARG;
        }
    p2_observe_cap(ch, zs_cap, hasher);
    for (u32 k = 0; k < nch; k++) chal.alpha[k] = ch.challenge();
    lap(1);

    // ---- quotient
    {
        p2_quotient_args a = {};
        p2_quotient_static_args(c, a);
        for (int k = 0; k < 4; k++) a.pih[k] = pih[k];
        a.ch = chal;
        {
            u32 n_pow = nch + nch * (npp + 1) + P.num_gate_constraints + 1;
            P2_PIN(c, tab, u32, (size_t)nch * n_pow * 6);        // pinned and alive until the next proof: the upload needs no wait
            for (u32 k = 0; k < nch; k++) {
                u64 v = 1;
                for (u32 i = 0; i < n_pow; i++) {
                    u32 *t6 = &tab[((size_t)k * n_pow + i) * 6];      // gl_limbs22 on the host
                    const u64 v2 = h_mul(v, 1ULL << 32);
                    t6[0] = (u32)(v & 0x3FFFFF), t6[1] = (u32)((v >> 22) & 0x3FFFFF), t6[2] = (u32)(v >> 44);
                    t6[3] = (u32)(v2 & 0x3FFFFF), t6[4] = (u32)((v2 >> 22) & 0x3FFFFF), t6[5] = (u32)(v2 >> 44);
                    v = h_mul(v, chal.alpha[k]);
                }
            }
            ZKLC_HIP(ctx, hipMemcpyAsync(c->d_apow, tab, (size_t)nch * n_pow * 6 * 4, hipMemcpyHostToDevice, st));
            for (u32 k = 0; k < nch; k++) a.apow[k] = c->d_apow + (size_t)k * n_pow * 6;
            for (u32 k = nch; k < P2_MAX_CH; k++) a.apow[k] = c->d_apow;
        }
        if (c->plan.nwaves) {
            hipLaunchKernelGGL(p2_quotient_fused_kernel, dim3(N / 64), dim3(64 * c->plan.nwaves), c->fused_lds, st, a, c->plan);
        } else {
            hipLaunchKernelGGL(p2_quotient_base_kernel, dim3((N + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st, a);
            static const bool one_per_launch = getenv("ZKLC_P2_GATE_LAUNCH") && !strcmp(getenv("ZKLC_P2_GATE_LAUNCH"), "single");   // A/B
            std::vector<bool> done(P.num_gates, false);
            for (u32 g = 0; g < P.num_gates; g++) {
                p2_gate_kernel_fn fn = p2_gate_kernel_of(c->gates[g].type);
                if (!fn || done[g]) continue;
                p2_gate_list list = {};
                for (u32 h = g; h < P.num_gates && list.n < (one_per_launch ? 1u : (u32)P2_GATE_LIST_MAX); h++)
                    if (!done[h] && c->gates[h].type == c->gates[g].type) {
                        list.idx[list.n++] = h;
                        done[h] = true;
                    }
                // A/B, opt-in: the one-pass evaluator of all U32AddMany variants measured the SAME 5.49 ms as the per-gate
                // launches on the Ed25519 circuit (profiles/r03v_*): see the comment at the kernel
                static const bool am_multi = getenv("ZKLC_P2_ADDMANY") && !strcmp(getenv("ZKLC_P2_ADDMANY"), "multi");
                // A/B: ZKLC_P2_ADDMANY=pergate keeps one launch per variant list; the default for several variants is the LDS tile
                static const bool am_pergate = getenv("ZKLC_P2_ADDMANY") && !strcmp(getenv("ZKLC_P2_ADDMANY"), "pergate");
                // debug (ZKLC_P2_ADDMANY=check): the per-gate evaluator and the LDS-tile kernel into zeroed scratch outputs, compared on
                // the host (every LDE point, both challenges); the proof itself then takes the per-gate path
                static const bool am_check = getenv("ZKLC_P2_ADDMANY") && !strcmp(getenv("ZKLC_P2_ADDMANY"), "check");
                if (am_check && c->gates[g].type == P2_U32_ADD_MANY && N >= 64) {
                    p2_amt_plan plan;
                    if (p2_amt_make_plan(c->gates.data(), list, plan)) {
                        u64 *s1 = nullptr, *s2 = nullptr;
                        const size_t words = (size_t)nch * N;
                        ZKLC_HIP(ctx, hipMalloc(&s1, words * 8));
                        ZKLC_HIP(ctx, hipMalloc(&s2, words * 8));
                        ZKLC_HIP(ctx, hipMemsetAsync(s1, 0, words * 8, st));
                        ZKLC_HIP(ctx, hipMemsetAsync(s2, 0, words * 8, st));
                        p2_quotient_args b1 = a, b2 = a;
                        b1.out = s1;
                        b2.out = s2;
                        hipLaunchKernelGGL(fn, dim3((N + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st, b1, list);
                        hipLaunchKernelGGL(p2_quotient_addmany_tile_kernel, dim3(N / 64), dim3(64 * P2_AMT_WAVES), 0, st, b2, list, plan);
                        std::vector<u64> h1(words), h2(words);
                        ZKLC_HIP(ctx, hipMemcpyAsync(h1.data(), s1, words * 8, hipMemcpyDeviceToHost, st));
                        ZKLC_HIP(ctx, hipMemcpyAsync(h2.data(), s2, words * 8, hipMemcpyDeviceToHost, st));
                        ZKLC_HIP(ctx, zklc_stream_wait(st));
                        size_t bad = 0;
                        for (size_t i = 0; i < words; i++) bad += h1[i] % GL_P != h2[i] % GL_P;
                        fprintf(stderr, "[zklc] addmany check: %u variants, limb columns [%u, %u): %zu of %zu values differ\n", list.n, plan.lo,
                                plan.hi, bad, words);
                        (void)hipFree(s1);
                        (void)hipFree(s2);
                    }
                }
                if (c->gates[g].type == P2_U32_ADD_MANY && !am_multi && !am_pergate && !am_check && N >= 64) {
                    p2_amt_plan plan;
                    if (p2_amt_make_plan(c->gates.data(), list, plan)) {
                        hipLaunchKernelGGL(p2_quotient_addmany_tile_kernel, dim3(N / 64), dim3(64 * P2_AMT_WAVES), 0, st, a, list, plan);
                        continue;
                    }
                }
                if (am_multi && c->gates[g].type == P2_U32_ADD_MANY && list.n >= 2 && list.n <= P2_AM_MAX)
                    fn = p2_quotient_addmany_multi_kernel;
                hipLaunchKernelGGL(fn, dim3((N + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st, a, list);
            }
        }
        ZKLC_HIP(ctx, hipGetLastError());
        // values on g<w_N> (bit-reversed) -> coefficients: inverse DIT, then undo the coset shift
        P2_RC(zklc_gl_ntt_dev(ctx, st, c->d_qv, c->lde_bits, nch, ZKLC_NTT_INVERSE | ZKLC_NTT_IN_BITREV, 0));
        hipLaunchKernelGGL(p2_scale_by_powers_kernel, dim3((N + 255) / 256, nch), dim3(256), 0, st, c->d_qv, h_inv(GL_GENERATOR), (u64)N);
        ZKLC_HIP(ctx, hipGetLastError());
        P2_RC(p2_commit_coeffs(c, st, c->quot, quot_cap));
    }
    p2_observe_cap(ch, quot_cap, hasher);
    gl2 zeta;
    zeta.a = ch.challenge();
    zeta.b = ch.challenge();
    lap(2);

    // ---- openings at zeta (all polynomials) and g*zeta (Zs)
    const u32 w0 = c->cs.width, w1 = c->wires.width, w2 = c->zs.width, w3 = c->quot.width;
    const u32 n_open = w0 + w1 + w2 + w3 + nch;
    P2_PIN(c, open, gl2, n_open);
    {
        hipLaunchKernelGGL(p2_ext_powers_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->d_zpow, zeta, (u64)n);
        const p2_batch *bs[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
        u32 off = 0;
        for (int k = 0; k < 4; k++) {
            hipLaunchKernelGGL(p2_eval_at_ext_kernel, dim3(bs[k]->width, P2_EVAL_SPLIT), dim3(P2_THREADS), 0, st,
                               (const u64 *)bs[k]->coeffs, n, (const gl2 *)c->d_zpow, (const u64 *)nullptr,
                               c->d_open_partial + (size_t)off * P2_EVAL_SPLIT);
            off += bs[k]->width;
        }
        hipLaunchKernelGGL(p2_eval_at_ext_kernel, dim3(nch, P2_EVAL_SPLIT), dim3(P2_THREADS), 0, st, (const u64 *)c->zs.coeffs, n,
                           (const gl2 *)c->d_zpow, (const u64 *)c->d_subgroup, c->d_open_partial + (size_t)off * P2_EVAL_SPLIT);
        hipLaunchKernelGGL(p2_eval_finish_kernel, dim3((n_open + 255) / 256), dim3(256), 0, st, (const gl2 *)c->d_open_partial, n_open,
                           c->d_open);
        ZKLC_HIP(ctx, hipGetLastError());
        ZKLC_HIP(ctx, zklc_readback_async(open, c->d_open, (size_t)n_open * sizeof(gl2), st));
        ZKLC_HIP(ctx, zklc_stream_wait(st));
    }
    // transcript order = FriOpenings: batch at zeta (constants, sigmas, wires, zs, partial products, quotient), then zs_next
    for (u32 i = 0; i < n_open; i++) {
        ch.observe(open[i].a);
        ch.observe(open[i].b);
    }
    gl2 fri_alpha;
    fri_alpha.a = ch.challenge();
    fri_alpha.b = ch.challenge();
    lap(3);

    // ---- FRI: batched quotient in evaluation form, commit phase
    std::vector<std::vector<uint8_t>> fri_caps(P.num_arities);
    std::vector<gl2> fri_betas(P.num_arities);
    u32 final_bits;
    {
        p2_fri_combine_args a = {};
        a.mats[0] = c->cs.lde;
        a.mats[1] = c->wires.lde;
        a.mats[2] = c->zs.lde;
        a.mats[3] = c->quot.lde;
        a.widths[0] = w0;
        a.widths[1] = w1;
        a.widths[2] = w2;
        a.widths[3] = w3;
        a.lde_bits = c->lde_bits;
        a.nch = nch;
        a.w_lde = h_root(c->lde_bits);
        a.alpha = fri_alpha;
        a.zeta = zeta;
        a.g_zeta = gl2_make(h_mul(zeta.a, h_root(P.degree_bits)), h_mul(zeta.b, h_root(P.degree_bits)));
        gl2 y0 = gl2_make(0, 0), y1 = gl2_make(0, 0);
        for (u32 i = w0 + w1 + w2 + w3; i-- > 0;) y0 = h2_add(h2_mul(y0, fri_alpha), open[i]);
        for (u32 i = nch; i-- > 0;) y1 = h2_add(h2_mul(y1, fri_alpha), open[w0 + w1 + w2 + w3 + i]);
        a.y0 = y0;
        a.y1 = y1;
        a.alpha_pow_nch = h2_pow(fri_alpha, nch);
        a.out = c->d_fri[0];
        a.apow = c->d_fri_apow;
        hipLaunchKernelGGL(p2_ext_pow_limbs_kernel, dim3((w0 + w1 + w2 + w3 + 255) / 256), dim3(256), 0, st, c->d_fri_apow, fri_alpha,
                           w0 + w1 + w2 + w3);
        hipLaunchKernelGGL(p2_fri_combine_kernel, dim3((N + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st, a);
        ZKLC_HIP(ctx, hipGetLastError());
        u32 bits = c->lde_bits;
        u64 shift = GL_GENERATOR;
        for (u32 r = 0; r < P.num_arities; r++) {
            u32 ab = P.arity_bits[r], A = 1u << ab;
            u32 leaves_bits = bits - ab;
            P2_RC(p2_merkle(c, st, (const u64 *)c->d_fri[r], 1, 2 * A, leaves_bits, 2 * A, c->d_fri_tree[r]));
            u32 cap_h = P.cap_height < leaves_bits ? P.cap_height : leaves_bits;
            fri_caps[r].resize((size_t)32 << cap_h);
            P2_PIN(c, h_fcap, uint8_t, fri_caps[r].size());
            ZKLC_HIP(ctx, zklc_readback_async(h_fcap, p2_tree_level(c->d_fri_tree[r], leaves_bits, leaves_bits - cap_h), fri_caps[r].size(), st));
            ZKLC_HIP(ctx, zklc_stream_wait(st));
            memcpy(fri_caps[r].data(), h_fcap, fri_caps[r].size());
            p2_observe_cap(ch, fri_caps[r], hasher);
            fri_betas[r].a = ch.challenge();
            fri_betas[r].b = ch.challenge();
            u32 chunks = 1u << leaves_bits;
            hipLaunchKernelGGL(p2_fri_fold_kernel, dim3((chunks + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st,
                               (const gl2 *)c->d_fri[r], c->d_fri[r + 1], bits, ab, shift, h_root(bits), h_inv(h_root(ab)),
                               h_inv(A), fri_betas[r]);
            ZKLC_HIP(ctx, hipGetLastError());
            shift = h_pow(shift, A);
            bits = leaves_bits;
        }
        final_bits = bits;
        u32 M = 1u << bits;
        hipLaunchKernelGGL(p2_fri_final_poly_kernel, dim3((M + P2_THREADS - 1) / P2_THREADS), dim3(P2_THREADS), 0, st,
                           (const gl2 *)c->d_fri[P.num_arities], c->d_final, bits, h_inv(shift), h_inv(h_root(bits)), h_inv(M));
        ZKLC_HIP(ctx, hipGetLastError());
    }
    const size_t final_n = (size_t)1 << final_bits;
    P2_PIN(c, final_poly, gl2, final_n);
    ZKLC_HIP(ctx, zklc_readback_async(final_poly, c->d_final, final_n * sizeof(gl2), st));
    ZKLC_HIP(ctx, zklc_stream_wait(st));
    const u32 final_len = 1u << (final_bits - P.rate_bits);
    for (size_t i = final_len; i < final_n; i++)
        if (final_poly[i].a || final_poly[i].b) {
            ctx->last_err = "plonky2: FRI final polynomial has a non-zero high coefficient (witness does not satisfy the circuit)";
            return ZKLC_ERR_INVALID_ARG;
        }
    for (u32 i = 0; i < final_len; i++) {
        ch.observe(final_poly[i].a);
        ch.observe(final_poly[i].b);
    }
    lap(4);

    // ---- proof of work
    u64 pow_witness;
    {
        p2_pow_args a = {};
        for (int i = 0; i < 12; i++) a.state[i] = ch.state[i];
        for (int i = 0; i < ch.n_in; i++) a.in[i] = ch.in[i];
        a.n_in = ch.n_in;
        a.pow_bits = P.proof_of_work_bits;
        a.found = c->d_found;
        const u64 batch = 1ULL << (P.proof_of_work_bits + 2 < 24 ? P.proof_of_work_bits + 2 : 24);  // ~98 % hit rate per launch
        P2_PIN(c, h_found, unsigned long long, 1);
        unsigned long long found = ~0ULL;
        for (u64 base = 0;; base += batch) {
            ZKLC_HIP(ctx, hipMemsetAsync(c->d_found, 0xFF, 8, st));
            a.base = base;
            hipLaunchKernelGGL(p2_pow_kernel, dim3((unsigned)(batch / P2_THREADS)), dim3(P2_THREADS), 0, st, a);
            ZKLC_HIP(ctx, hipGetLastError());
            ZKLC_HIP(ctx, zklc_readback_async(h_found, c->d_found, 8, st));
            ZKLC_HIP(ctx, zklc_stream_wait(st));
            found = *h_found;
            if (found != ~0ULL) break;
            if (base > (1ULL << 50)) return ZKLC_ERR_INVALID_ARG;
        }
        pow_witness = found;
        ch.observe(pow_witness);
        u64 resp = ch.challenge();
        if (resp >> (64 - P.proof_of_work_bits)) {
            ctx->last_err = "plonky2: proof-of-work response mismatch between the kernel and the host transcript";
            return ZKLC_ERR_HIP;
        }
    }
    lap(5);

    // ---- query rounds: gather leaves, Merkle paths and FRI cosets
    const size_t n_qwords = L.per_round_words * P.num_query_rounds;
    P2_PIN(c, qwords, u64, n_qwords);
    {
        std::vector<p2_gather> gs;
        const p2_batch *bs[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
        size_t dst = 0;
        for (u32 q = 0; q < P.num_query_rounds; q++) {
            u64 x_index = ch.challenge() & (N - 1);
            for (int k = 0; k < 4; k++) {
                gs.push_back({bs[k]->lde + x_index, (u64)N, bs[k]->width, (u32)dst});
                dst += bs[k]->width;
                for (u32 l = 0; l < L.depth0; l++) {
                    gs.push_back({p2_tree_level(bs[k]->tree, c->lde_bits, l) + 4 * ((x_index >> l) ^ 1), 1, 4, (u32)dst});
                    dst += 4;
                }
            }
            u32 bits = c->lde_bits;
            u64 idx = x_index;
            for (u32 r = 0; r < P.num_arities; r++) {
                u32 ab = P.arity_bits[r];
                idx >>= ab;
                bits -= ab;
                gs.push_back({(const u64 *)(c->d_fri[r] + (idx << ab)), 1, 2u << ab, (u32)dst});
                dst += 2u << ab;
                u32 cap_h = P.cap_height < bits ? P.cap_height : bits;
                for (u32 l = 0; l < bits - cap_h; l++) {
                    gs.push_back({p2_tree_level(c->d_fri_tree[r], bits, l) + 4 * ((idx >> l) ^ 1), 1, 4, (u32)dst});
                    dst += 4;
                }
            }
        }
        if (dst != n_qwords) return ZKLC_ERR_INVALID_ARG;
        if (c->gather_cap < gs.size()) {
            P2_ALLOC(c, c->d_gather, gs.size() * sizeof(p2_gather));
            c->gather_cap = gs.size();
        }
        if (c->gather_out_words < n_qwords) {
            P2_ALLOC(c, c->d_gather_out, n_qwords * 8);
            c->gather_out_words = n_qwords;
        }
        P2_PIN(c, h_gs, p2_gather, gs.size());
        memcpy(h_gs, gs.data(), gs.size() * sizeof(p2_gather));
        ZKLC_HIP(ctx, hipMemcpyAsync(c->d_gather, h_gs, gs.size() * sizeof(p2_gather), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(p2_gather_kernel, dim3((unsigned)gs.size()), dim3(64), 0, st, (const p2_gather *)c->d_gather, (u32)gs.size(),
                           c->d_gather_out);
        ZKLC_HIP(ctx, hipGetLastError());
        ZKLC_HIP(ctx, zklc_readback_async(qwords, c->d_gather_out, n_qwords * 8, st));
        ZKLC_HIP(ctx, zklc_stream_wait(st));
    }

    // ---- serialise
    uint8_t *o = proof_out;
    auto put = [&](const void *p, size_t nbytes) {
        memcpy(o, p, nbytes);
        o += nbytes;
    };
    put(wires_cap.data(), wires_cap.size());
    put(zs_cap.data(), zs_cap.size());
    put(quot_cap.data(), quot_cap.size());
    // openings: constants, sigmas, wires, zs, zs_next, partial products, quotient
    put(open, (size_t)(w0 + w1 + nch) * 16);
    put(open + (w0 + w1 + w2 + w3), (size_t)nch * 16);
    put(open + (w0 + w1 + nch), (size_t)(w2 - nch + w3) * 16);
    for (u32 r = 0; r < P.num_arities; r++) put(fri_caps[r].data(), fri_caps[r].size());
    {
        const u64 *w = qwords;
        u32 widths[4] = {w0, w1, w2, w3};
        for (u32 q = 0; q < P.num_query_rounds; q++) {
            for (int k = 0; k < 4; k++) {
                put(w, 8 * (size_t)widths[k]);
                w += widths[k];
                *o++ = (uint8_t)L.depth0;
                put(w, 32 * (size_t)L.depth0);
                w += 4 * (size_t)L.depth0;
            }
            u32 bits = c->lde_bits;
            for (u32 r = 0; r < P.num_arities; r++) {
                u32 ab = P.arity_bits[r];
                bits -= ab;
                put(w, 16 * ((size_t)1 << ab));
                w += 2 * ((size_t)1 << ab);
                u32 cap_h = P.cap_height < bits ? P.cap_height : bits;
                *o++ = (uint8_t)(bits - cap_h);
                put(w, 32 * (size_t)(bits - cap_h));
                w += 4 * (size_t)(bits - cap_h);
            }
        }
    }
    put(final_poly, (size_t)final_len * 16);
    put(&pow_witness, 8);
    u64 npi = P.num_public_inputs;
    put(&npi, 8);
    put(public_inputs, 8 * (size_t)npi);
    if ((size_t)(o - proof_out) != L.bytes) return ZKLC_ERR_INVALID_ARG;
    *proof_len = L.bytes;
    lap(6);
    c->timings[7] = p2_now_ms() - t0;
    for (u32 k = 0; k < nch; k++) c->last_challenges.push_back(chal.beta[k]);
    for (u32 k = 0; k < nch; k++) c->last_challenges.push_back(chal.gamma[k]);
    for (u32 k = 0; k < nch; k++) c->last_challenges.push_back(chal.alpha[k]);
    c->last_challenges.push_back(zeta.a);
    c->last_challenges.push_back(zeta.b);
    c->last_challenges.push_back(fri_alpha.a);
    c->last_challenges.push_back(fri_alpha.b);
    for (u32 r = 0; r < P.num_arities; r++) {
        c->last_challenges.push_back(fri_betas[r].a);
        c->last_challenges.push_back(fri_betas[r].b);
    }
    return ZKLC_OK;
}

extern "C" int32_t zklc_plonky2_prove(zklc_ctx *ctx, zklc_plonky2_circuit *c, const uint64_t *wires, const uint64_t *public_inputs,
                                      uint8_t *proof_out, uint64_t proof_cap, uint64_t *proof_len) {
    if (!ctx || !c || c->ctx != ctx || !wires) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t bytes = (size_t)c->P.num_wires * c->n * 8;
    ZKLC_HIP(ctx, hipMemcpyAsync(c->d_wire_vals, wires, bytes, hipMemcpyHostToDevice, ctx->stream));
    return zklc_plonky2_prove_dev(ctx, ctx->stream, c, c->d_wire_vals, public_inputs, proof_out, proof_cap, proof_len);
}
