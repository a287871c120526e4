// BN254 G1 and G2 multi-scalar multiplication (Pippenger bucket method) for gfx950 + C ABI.
//
// Replaces gnark-crypto's `bn254.G1Affine.MultiExp` / `bn254.G2Affine.MultiExp` (un-vendored,
// gnark-plonky2-verifier/go.mod:9) under `groth16.Prove`
// (gnark-plonky2-verifier/cmd/web-api.go:77, tests/prover_test.go:64).  Every kernel is a template over the
// coordinate field (Fp for G1, Fp2 for G2; bn254_ec.cuh) -- the pipeline is identical.  What one lane does lives in
// bn254_msm_lane.cuh (shared with tests/hostsim); this file is the parallel structure around it.
//
// Pipeline (all on the device, no host round trips, NO global atomics):
//   1. recode: signed c-bit digits of every scalar, 16 bits each, window-major (a point at infinity gets zero digits)
//   2. counting sort by (window, bucket) on (chunk, window) tiles, one workgroup per tile with the window's 2^(c-1) counters in
//      LDS (128 KiB at c = 16): histogram per tile -> prefix over the chunks + bucket totals -> exclusive scan of the totals ->
//      scatter of (point index, sign) entries through LDS cursors.  Round 1 did both passes with one global atomic per
//      (scalar, window): 5.7 of the 27 ms of a 2^22 MSM.
//   3. bucket accumulation over SLICES of the sorted entries: a lane adds exactly MSM_SLICE consecutive entries (extended-Jacobian
//      mixed additions, the gathers software-pipelined one iteration ahead), whatever buckets they belong to -- whole buckets are
//      stored, the pieces cut by slice boundaries are added by a one-lane-per-bucket pass (a workgroup for buckets cut into many
//      slices: skewed scalars, e.g. many equal to 1).  Every lane does the same work for ANY digit distribution.
//   4. bucket reduction: per window, segments of MSM_SEG buckets -> running-sum trick + small scalar multiple (one lane per
//      segment), then an LDS tree over the segments
//   5. 2^(c w) * window_w by doublings (one lane per window), tree sum, affine output
// Curve additions are order-independent as group elements, so the (arbitrary) order of the LDS atomics in step 2 never changes
// the affine result.
#include "bn254_msm_lane.cuh"
#include "zklc_internal.h"
#include <stdlib.h>

#define MSM_SORT_THREADS 1024
#define MSM_SLICE_WAVES_G1 2

// ---- 1. recode
template <int AFF>
__global__ void __launch_bounds__(256)
msm_recode_kernel(const u64 *__restrict__ points, const u64 *__restrict__ scalars, msm_plan pl, unsigned short *__restrict__ dig) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pl.n_pad) return;
    bool live = i < pl.n && !msm_point_is_inf<AFF>(points, i);
    u32 sw[8];
    if (live) msm_load_scalar(scalars, i, sw);
    u32 carry = 0;
    for (u32 w = 0; w < pl.windows; w++) {
        int d = live ? msm_digit(sw, w, pl.c, carry) : 0;
        dig[(size_t)w * pl.n_pad + i] = (unsigned short)msm_digit_code(d);
    }
}

// ---- 1'. the endomorphism split (G1): one lane per POINT: k -> (k1, k2), the digits of both magnitudes, and the two converted point
// records (x, +-y) / (beta x, +-y) with the signs folded into y (bn254_msm_lane.cuh: msm_glv_split, msm_convert_point_glv)
template <bool PK>
__global__ void __launch_bounds__(256)
msm_glv_prepare_kernel(const u64 *__restrict__ points, const u64 *__restrict__ scalars, msm_plan pl, unsigned short *__restrict__ dig,
                       i32 *__restrict__ cpoints) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (pl.n + i < pl.n_pad)                              // the (at most seven) pad items of every digit row
        for (u32 w = 0; w < pl.windows; w++) dig[(size_t)w * pl.n_pad + pl.n + i] = 0;
    if (i >= pl.n_pts) return;
    const bool live = !msm_point_is_inf<8>(points, i);
    u32 sw[8], m1[8], m2[8], neg1 = 0, neg2 = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) sw[k] = m1[k] = m2[k] = 0;
    if (live) {
        msm_load_scalar(scalars, i, sw);
        msm_glv_split(sw, m1, neg1, m2, neg2);
    }
    u32 carry1 = 0, carry2 = 0;
    for (u32 w = 0; w < pl.windows; w++) {
        int d1 = live ? msm_digit(m1, w, pl.c, carry1) : 0, d2 = live ? msm_digit(m2, w, pl.c, carry2) : 0;
        dig[(size_t)w * pl.n_pad + i] = (unsigned short)msm_digit_code(d1);
        dig[(size_t)w * pl.n_pad + pl.n_pts + i] = (unsigned short)msm_digit_code(d2);
    }
    const int W = msm_rec<FpField, PK>::WORDS;
    msm_convert_point_glv<FpField, PK>(cpoints + (size_t)i * W, cpoints + ((size_t)pl.n_pts + i) * W, points, i, neg1, neg2);
}

// eight digit codes of one lane (16 bytes)
ZKLC_D void msm_codes8(const unsigned short *row, u32 i, u32 *code) {
    uint4 v = *reinterpret_cast<const uint4 *>(row + i);
    code[0] = v.x & 0xffff; code[1] = v.x >> 16; code[2] = v.y & 0xffff; code[3] = v.y >> 16;
    code[4] = v.z & 0xffff; code[5] = v.z >> 16; code[6] = v.w & 0xffff; code[7] = v.w >> 16;
}

// ---- 2a. histogram of tile (chunk k = blockIdx.x, window w = blockIdx.y) in LDS -> cnt[w][k][bucket]
__global__ void __launch_bounds__(MSM_SORT_THREADS)
msm_hist_kernel(const unsigned short *__restrict__ dig, msm_plan pl, u32 *__restrict__ cnt) {
    extern __shared__ u32 msm_lds[];
    const u32 bpw = pl.buckets_per_window, k = blockIdx.x, w = blockIdx.y;
    for (u32 b = threadIdx.x; b < bpw; b += MSM_SORT_THREADS) msm_lds[b] = 0;
    __syncthreads();
    const unsigned short *row = dig + (size_t)w * pl.n_pad;
    u32 lo = k * pl.chunk_len, hi = lo + pl.chunk_len < pl.n_pad ? lo + pl.chunk_len : pl.n_pad;
    for (u32 i = lo + 8 * threadIdx.x; i < hi; i += 8 * MSM_SORT_THREADS) {
        u32 code[8];
        msm_codes8(row, i, code);
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (code[j]) {
                u32 neg;
                atomicAdd(&msm_lds[msm_code_bucket(code[j], neg)], 1u);
            }
    }
    __syncthreads();
    u32 *out = cnt + ((size_t)w * pl.chunks + k) * bpw;
    for (u32 b = threadIdx.x; b < bpw; b += MSM_SORT_THREADS) out[b] = msm_lds[b];
}

// ---- 2b. per bucket: exclusive prefix of its tile counts over the chunks (in place) and its total
__global__ void __launch_bounds__(256) msm_totals_kernel(u32 *__restrict__ cnt, msm_plan pl, u32 *__restrict__ totals) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= pl.total_buckets) return;
    u32 w = key / pl.buckets_per_window, b = key % pl.buckets_per_window;
    u32 run = 0;
    for (u32 k = 0; k < pl.chunks; k++) {
        u32 *p = cnt + ((size_t)w * pl.chunks + k) * pl.buckets_per_window + b;
        u32 t = *p;
        *p = run;
        run += t;
    }
    totals[key] = run;
}

// ---- 2c. scatter of tile (k, w): LDS cursors start at (bucket offset + prefix of this tile)
__global__ void __launch_bounds__(MSM_SORT_THREADS)
msm_scatter_kernel(const unsigned short *__restrict__ dig, msm_plan pl, const u32 *__restrict__ cnt, const u32 *__restrict__ offsets,
                   u32 *__restrict__ entries, u32 fixed_n) {
    // fixed_n != 0 (fixed-base form): ONE bucket set for all windows -- the offsets are indexed by the bucket alone -- and an entry
    // names row (w, i) of the table of 2^(c w) P_i: index w * fixed_n + i
    extern __shared__ u32 msm_lds[];
    const u32 bpw = pl.buckets_per_window, k = blockIdx.x, w = blockIdx.y;
    const u32 *pre = cnt + ((size_t)w * pl.chunks + k) * bpw;
    const u32 row_base = w * fixed_n;
    for (u32 b = threadIdx.x; b < bpw; b += MSM_SORT_THREADS) msm_lds[b] = offsets[(fixed_n ? 0 : (size_t)w * bpw) + b] + pre[b];
    __syncthreads();
    const unsigned short *row = dig + (size_t)w * pl.n_pad;
    u32 lo = k * pl.chunk_len, hi = lo + pl.chunk_len < pl.n_pad ? lo + pl.chunk_len : pl.n_pad;
    for (u32 i = lo + 8 * threadIdx.x; i < hi; i += 8 * MSM_SORT_THREADS) {
        u32 code[8];
        msm_codes8(row, i, code);
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (code[j]) {
                u32 neg;
                u32 b = msm_code_bucket(code[j], neg);
                u32 pos = atomicAdd(&msm_lds[b], 1u);
                entries[pos] = ((row_base + i + j) << 1) | neg;
            }
    }
}

// ---- exclusive scan of `n` u32 (n <= 2^24): 1024 items per block, block sums scanned by one block
#define SCAN_ITEMS 1024
__global__ void __launch_bounds__(256) msm_scan_block_kernel(const u32 *in, u32 *out, u32 *block_sums, u32 n) {
    __shared__ u32 tmp[SCAN_ITEMS];
    u32 base = blockIdx.x * SCAN_ITEMS;
    for (u32 k = threadIdx.x; k < SCAN_ITEMS; k += 256) tmp[k] = base + k < n ? in[base + k] : 0;
    __syncthreads();
    // each thread scans 4 consecutive items, then a block scan over the 256 partial sums
    u32 a0 = tmp[4 * threadIdx.x], a1 = tmp[4 * threadIdx.x + 1], a2 = tmp[4 * threadIdx.x + 2], a3 = tmp[4 * threadIdx.x + 3];
    u32 s = a0 + a1 + a2 + a3;
    __shared__ u32 part[256];
    part[threadIdx.x] = s;
    __syncthreads();
    for (u32 off = 1; off < 256; off <<= 1) {
        u32 v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    u32 excl = part[threadIdx.x] - s;
    u32 o = base + 4 * threadIdx.x;
    if (o < n) out[o] = excl;
    if (o + 1 < n) out[o + 1] = excl + a0;
    if (o + 2 < n) out[o + 2] = excl + a0 + a1;
    if (o + 3 < n) out[o + 3] = excl + a0 + a1 + a2;
    if (threadIdx.x == 255) block_sums[blockIdx.x] = part[255];
}
__global__ void __launch_bounds__(256) msm_scan_sums_kernel(u32 *block_sums, u32 nblocks) {
    // single block, sequential over chunks of 256 (nblocks <= 16384)
    __shared__ u32 part[256];
    __shared__ u32 running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (u32 base = 0; base < nblocks; base += 256) {
        u32 idx = base + threadIdx.x;
        u32 v = idx < nblocks ? block_sums[idx] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (u32 off = 1; off < 256; off <<= 1) {
            u32 t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        u32 excl = running + part[threadIdx.x] - v;
        if (idx < nblocks) block_sums[idx] = excl;
        __syncthreads();
        if (threadIdx.x == 255) running += part[255];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) msm_scan_add_kernel(u32 *out, const u32 *block_sums, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += block_sums[i / SCAN_ITEMS];
}

// ---- 3. bucket accumulation over SLICES of the sorted entries (bn254_msm_lane.cuh: msm_slice_lane), template over the coordinate
// field F (FpField = G1, Fp2Field = G2); WAVES = the occupancy the register allocation is asked to keep
template <class F, bool PK>
__global__ void __launch_bounds__(256) msm_convert_kernel(const u64 *__restrict__ points, u32 n, i32 *__restrict__ cpoints) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) msm_convert_point<F, PK>(cpoints + (size_t)i * msm_rec<F, PK>::WORDS, points, i);
}

#define MSM_SLICE_BLOCK 256
template <class F, int WAVES, bool PK>
__global__ void __launch_bounds__(MSM_SLICE_BLOCK, WAVES)
msm_slice_kernel(const i32 *__restrict__ cpoints, const u32 *__restrict__ entries, const u32 *__restrict__ offsets,
                 const u32 *__restrict__ counts, msm_plan pl, i32 *__restrict__ buckets, i32 *__restrict__ partials) {
    u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    u32 T = pl.total_buckets;
    u32 E = offsets[T - 1] + counts[T - 1];
    msm_slice_lane<F, PK>(cpoints, entries, offsets, T, E, lane, buckets, partials);
}

// one lane per bucket: the sum of the partials of a bucket cut by slice boundaries (usually two of them); buckets cut into more
// than MSM_COMBINE_SERIAL slices (skewed scalars: many equal digits) go on the list of the workgroup-per-bucket kernel
template <class F>
__global__ void __launch_bounds__(64)
msm_combine_kernel(const u32 *__restrict__ offsets, const u32 *__restrict__ counts, msm_plan pl, const i32 *__restrict__ partials,
                   i32 *__restrict__ buckets, u32 *__restrict__ heavy_list, u32 *__restrict__ heavy_count) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= pl.total_buckets) return;
    u32 la, lb, which;
    if (!msm_combine_span<F>(offsets, counts, key, la, lb, which)) return;
    if (lb - la + 1 > MSM_COMBINE_SERIAL) {
        u32 slot = atomicAdd(heavy_count, 1u);
        if (slot < MSM_MAX_HEAVY) {
            heavy_list[slot] = key;
            return;
        }
        // list full: fall through and do it here (slow but correct)
    }
    msm_combine_lane<F>(offsets, counts, key, partials, buckets);
}

// LDS tree reduction of one XYZZ point per thread (BLOCK threads); result in thread 0
template <class F>
ZKLC_D ec_xyzz<F> msm_block_reduce(ec_xyzz<F> acc, i32 *lds /* BLOCK * XYZZ words */) {
    const int XY = msm_cfg<F>::XYZZ;
    for (u32 stride = msm_cfg<F>::BLOCK / 2; stride >= 1; stride >>= 1) {
        if (threadIdx.x >= stride && threadIdx.x < 2 * stride) msm_store_xyzz<F>(lds + (threadIdx.x - stride) * XY, acc);
        __syncthreads();
        if (threadIdx.x < stride) acc = ec_add(acc, msm_load_xyzz<F>(lds + threadIdx.x * XY));
        __syncthreads();
    }
    return acc;
}

template <class F>
__global__ void __launch_bounds__(msm_cfg<F>::BLOCK)
msm_heavy_combine_kernel(const u32 *__restrict__ offsets, const u32 *__restrict__ counts, const i32 *__restrict__ partials,
                         i32 *__restrict__ buckets, const u32 *__restrict__ heavy_list, const u32 *__restrict__ heavy_count) {
    const int XY = msm_cfg<F>::XYZZ;
    __shared__ i32 lds[msm_cfg<F>::BLOCK * msm_cfg<F>::XYZZ];
    u32 nheavy = *heavy_count;
    if (nheavy > MSM_MAX_HEAVY) nheavy = MSM_MAX_HEAVY;
    if (blockIdx.x >= nheavy) return;
    u32 key = heavy_list[blockIdx.x];
    u32 la, lb, which;
    msm_combine_span<F>(offsets, counts, key, la, lb, which);
    ec_xyzz<F> acc = ec_infinity<F>();
    for (u32 j = la + threadIdx.x; j <= lb; j += msm_cfg<F>::BLOCK)
        acc = ec_add(acc, msm_load_xyzz<F>(partials + ((size_t)2 * j + (j == la ? which : 0u)) * XY));
    acc = msm_block_reduce<F>(acc, lds);
    if (threadIdx.x == 0) msm_store_xyzz<F>(buckets + (size_t)key * XY, acc);
}

// ---- 4. one lane per segment of MSM_SEG buckets
template <class F>
__global__ void __launch_bounds__(64) msm_segment_kernel(const i32 *__restrict__ buckets, msm_plan pl, i32 *__restrict__ seg_out) {
    u32 seg_per_window = (pl.buckets_per_window + MSM_SEG - 1) / MSM_SEG;
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= seg_per_window * pl.windows) return;
    ec_xyzz<F> T = msm_segment_lane<F>(buckets, pl, s / seg_per_window, s % seg_per_window);
    msm_store_xyzz<F>(seg_out + (size_t)s * msm_cfg<F>::XYZZ, T);
}

// one workgroup per window: sum of its segment results
template <class F>
__global__ void __launch_bounds__(msm_cfg<F>::BLOCK) msm_window_kernel(const i32 *__restrict__ seg_out, msm_plan pl, i32 *__restrict__ win_out) {
    const int XY = msm_cfg<F>::XYZZ;
    __shared__ i32 lds[msm_cfg<F>::BLOCK * msm_cfg<F>::XYZZ];
    u32 seg_per_window = (pl.buckets_per_window + MSM_SEG - 1) / MSM_SEG;
    u32 w = blockIdx.x;
    ec_xyzz<F> acc = ec_infinity<F>();
    for (u32 s = threadIdx.x; s < seg_per_window; s += msm_cfg<F>::BLOCK)
        acc = ec_add(acc, msm_load_xyzz<F>(seg_out + ((size_t)w * seg_per_window + s) * XY));
    acc = msm_block_reduce<F>(acc, lds);
    if (threadIdx.x == 0) msm_store_xyzz<F>(win_out + (size_t)w * XY, acc);
}

// quad broadcast of a field element (DPP quad_perm [SRC, SRC, SRC, SRC] on every limb)
template <int SRC>
ZKLC_D fp msm_quad_bcast(const fp &v) {
    fp r;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int k = 0; k < 10; k++) r.v[k] = __builtin_amdgcn_mov_dpp(v.v[k], SRC * 0x55, 0xf, 0xf, false);
#else
    r = v;
#endif
    return r;
}
template <int SRC>
ZKLC_D fp2 msm_quad_bcast(const fp2 &v) {
    fp2 r;
    r.c0 = msm_quad_bcast<SRC>(v.c0);
    r.c1 = msm_quad_bcast<SRC>(v.c1);
    return r;
}
// one doubling by the four lanes of a quad (bn254_msm_lane.cuh: ecq_stage*); every lane holds the point before and after
template <class F>
ZKLC_D void msm_double_quad(ec_xyzz<F> &p, u32 role) {
    typedef typename F::T T;
    T r1 = ecq_stage1<F>(p, role);
    T V = msm_quad_bcast<0>(r1), XX = msm_quad_bcast<1>(r1);
    T r2 = ecq_stage2<F>(p, V, XX, role);
    T W = msm_quad_bcast<0>(r2), S = msm_quad_bcast<1>(r2), ZZ3 = msm_quad_bcast<2>(r2), MM = msm_quad_bcast<3>(r2);
    T M = F::add(F::dbl(XX), XX);
    T X3 = F::sub(MM, F::dbl(S));
    T r3 = ecq_stage3<F>(p, W, S, M, X3, role);
    T ZZZ3 = msm_quad_bcast<0>(r3), WY = msm_quad_bcast<1>(r3), MS = msm_quad_bcast<2>(r3);
    p.X = X3;
    p.Y = F::sub(MS, WY);
    p.ZZ = ZZ3;
    p.ZZZ = ZZZ3;
}

// result = sum_w 2^(c w) W_w ; affine output in gnark Montgomery words + infinity flag.  A quad of lanes per window shares every
// doubling (msm_double_quad: three multiplication-times instead of nine on the 240-doubling serial tail).
template <class F>
__global__ void __launch_bounds__(msm_cfg<F>::BLOCK) msm_final_kernel(const i32 *__restrict__ win_out, msm_plan pl, u64 *__restrict__ out_affine, u32 *__restrict__ out_inf) {
    const int XY = msm_cfg<F>::XYZZ;
    __shared__ i32 lds[msm_cfg<F>::BLOCK * msm_cfg<F>::XYZZ];
    ec_xyzz<F> acc = ec_infinity<F>();
    if (pl.windows == 1) {              // the fixed-base form: one window, nothing to double or to add -- only the affine conversion
        if (threadIdx.x == 0) {
            acc = msm_load_xyzz<F>(win_out);
            const int W = 2 * msm_cfg<F>::AFF;
            u32 o[W];
            u32 inf = ec_to_affine_gnark(o, acc);
            for (int k = 0; k < W / 2; k++) out_affine[k] = (u64)o[2 * k] | ((u64)o[2 * k + 1] << 32);
            *out_inf = inf;
        }
        return;
    }
    // a quad of lanes per window shares every doubling (G1 and G2); with more windows than quads (tiny inputs) one lane per window
    const bool quads = 4 * pl.windows <= (u32)msm_cfg<F>::BLOCK;
    if (quads && threadIdx.x < 4 * pl.windows) {
        u32 w = threadIdx.x >> 2, role = threadIdx.x & 3;
        ec_xyzz<F> p = msm_load_xyzz<F>(win_out + (size_t)w * XY);
        u32 dbl = pl.c * w;
#pragma unroll 1
        for (u32 k = 0; k < dbl; k++) msm_double_quad<F>(p, role);
        if (role == 0) acc = p;
    }
    if (!quads && threadIdx.x < pl.windows) {
        acc = msm_load_xyzz<F>(win_out + (size_t)threadIdx.x * XY);
        u32 dbl = pl.c * threadIdx.x;
        for (u32 k = 0; k < dbl; k++) acc = ec_double(acc);
    }
    acc = msm_block_reduce<F>(acc, lds);
    if (threadIdx.x == 0) {
        const int W = 2 * msm_cfg<F>::AFF;  // u32 words of the affine output
        u32 o[W];
        u32 inf = ec_to_affine_gnark(o, acc);
        for (int k = 0; k < W / 2; k++) out_affine[k] = (u64)o[2 * k] | ((u64)o[2 * k + 1] << 32);
        *out_inf = inf;
    }
}

// ---------------------------------------------------------------- fixed-base form (round 5)
// A Groth16 proving key is a FIXED set of bases (gnark-plonky2-verifier/cmd/web-api.go:77 multiplies the same pk.G1.A / B / K / Z
// by every proof's witness): row w of the table holds 2^(c w) P_i as packed affine records, so the digit of window w adds a table
// point straight into ONE bucket set shared by all windows -- no 2^(c w) doublings at the end (the 240-doubling serial chain, 0.9 ms
// of a 2^22 multi-exponentiation) and one bucket reduction instead of sixteen.  Same additions per point, same result (the affine
// output is canonical: tests compare it with the plain form bit for bit); 16 x the memory of the bases, built once.
#define MSM_TABLE_HEADER_WORDS 64          // 256 bytes: [magic, n, c, windows, words per record]
#define MSM_TABLE_MAGIC 0x7a6b6c54u
template <class F>
__global__ void __launch_bounds__(64) msm_fixed_table_kernel(const u64 *__restrict__ points, u32 n, u32 c, u32 windows, i32 *__restrict__ table) {
    typedef typename F::T T;
    const int W = msm_rec<F, true>::WORDS;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (msm_point_is_inf<msm_cfg<F>::AFF>(points, i)) {          // all-zero records: the recoding gives such a point no digit
        for (u32 w = 0; w < windows; w++)
            for (int k = 0; k < W; k++) table[((size_t)w * n + i) * W + k] = 0;
        return;
    }
    T x, y;
    msm_load_point<F>(points, i, x, y);
    x = F::reduce(x);
    y = F::reduce(y);
#pragma unroll 1
    for (u32 w = 0; w < windows; w++) {
        u32 *dst = reinterpret_cast<u32 *>(table + ((size_t)w * n + i) * W);
        F::pack(dst, x);
        F::pack(dst + F::PACKW, y);
        if (w + 1 == windows) break;
        ec_xyzz<F> acc;
        acc.X = x;
        acc.Y = y;
        acc.ZZ = acc.ZZZ = F::one();
#pragma unroll 1
        for (u32 k = 0; k < c; k++) acc = ec_double(acc);
        T inv = F::inv(F::mul(acc.ZZ, acc.ZZZ));                // a point of prime order never doubles to infinity
        x = F::reduce(F::mul(acc.X, F::mul(inv, acc.ZZZ)));
        y = F::reduce(F::mul(acc.Y, F::mul(inv, acc.ZZ)));
    }
}
// digits of the scalars; a base is skipped when its table record is all zeros (= the point at infinity)
template <class F>
__global__ void __launch_bounds__(256)
msm_recode_fixed_kernel(const i32 *__restrict__ table, const u64 *__restrict__ scalars, msm_plan pl, unsigned short *__restrict__ dig) {
    const int W = msm_rec<F, true>::WORDS;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pl.n_pad) return;
    bool live = i < pl.n;
    if (live) {
        i32 acc = 0;
        for (int k = 0; k < W; k++) acc |= table[(size_t)i * W + k];
        live = acc != 0;
    }
    u32 sw[8];
    if (live) msm_load_scalar(scalars, i, sw);
    u32 carry = 0;
    for (u32 w = 0; w < pl.windows; w++) {
        int d = live ? msm_digit(sw, w, pl.c, carry) : 0;
        dig[(size_t)w * pl.n_pad + i] = (unsigned short)msm_digit_code(d);
    }
}

// ---------------------------------------------------------------- host
// A/B switch ZKLC_MSM_GLV=0: no endomorphism split (G1: the plan of rounds 1-3)
static bool msm_glv_enabled() {
    static const bool on = !(getenv("ZKLC_MSM_GLV") && getenv("ZKLC_MSM_GLV")[0] == '0');
    return on;
}
template <class F>
static uint64_t msm_workspace_bytes_plan(const msm_plan &pl) {
    const uint64_t XB = msm_cfg<F>::XYZZ * 4;
    u32 seg_per_window = (pl.buckets_per_window + MSM_SEG - 1) / MSM_SEG;
    uint64_t total = pl.total_buckets, n = pl.n, b = 0;
    b += total * 4 * 2;                                 // totals (bucket sizes), offsets
    b += (total / SCAN_ITEMS + 2) * 4;                  // scan block sums
    b += total * pl.chunks * 4;                         // per-tile counts / prefixes
    b += (uint64_t)pl.n_pad * pl.windows * 2;           // digit codes
    b += n * pl.windows * 4 + 4;                        // entries
    b += total * XB;                                    // buckets
    b += n * 2 * F::LIMBS * 4;                          // converted points
    b += ((n * pl.windows + MSM_SLICE - 1) / MSM_SLICE + 1) * 2 * XB;   // slice partials
    b += (uint64_t)seg_per_window * pl.windows * XB;    // segment sums
    b += (uint64_t)pl.windows * XB;                     // window sums
    b += (MSM_MAX_HEAVY + 4) * 4;                       // heavy list + counter
    b += 256 * 16;                                      // alignment slack
    return b;
}
template <class F>
static uint64_t msm_workspace_bytes(uint64_t n) {       // enough for either plan (the A/B switch may differ between the two calls)
    uint64_t a = msm_workspace_bytes_plan<F>(msm_make_plan(n, false));
    if (F::LIMBS == 10) {
        uint64_t b = msm_workspace_bytes_plan<F>(msm_make_plan(n, true));
        a = b > a ? b : a;
    }
    return a;
}

// the sort kernels keep a whole window's counters in LDS: up to 128 KiB of dynamic LDS, above the default 64 KiB limit
static hipError_t msm_sort_lds_attr() {
    return zklc_once_per_device([] {
        hipError_t e = hipFuncSetAttribute((const void *)msm_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute((const void *)msm_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
}

template <class F>
static int32_t msm_run_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, const uint64_t *d_scalars, uint64_t n,
                           uint64_t *d_out_affine, uint32_t *d_out_inf, void *d_workspace, uint64_t workspace_bytes,
                           const i32 *fixed_table = nullptr) {
    // fixed_table != nullptr: the fixed-base form -- d_points is not read, the records come from the table (rows of 2^(c w) P_i)
    const bool fixed = fixed_table != nullptr;
    const int AFF = msm_cfg<F>::AFF;
    const size_t XB = msm_cfg<F>::XYZZ * 4;
    if (!ctx || !d_out_affine || !d_out_inf || (n && ((!fixed && !d_points) || !d_scalars)) || n >= (1ULL << 30)) return ZKLC_ERR_INVALID_ARG;
    if (((uintptr_t)d_points | (uintptr_t)d_scalars) & 15) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    ZKLC_HIP(ctx, msm_sort_lds_attr());
    hipStream_t st = zklc_pick_stream(ctx, stream);
    msm_plan pl = msm_make_plan(n, !fixed && F::LIMBS == 10 && msm_glv_enabled());
    if (fixed && (uint64_t)pl.n * pl.windows >= (1ULL << 31)) return ZKLC_ERR_INVALID_ARG;       // entry = (row index << 1) | sign
    // the view of the kernels behind the sort: ONE window whose tiles are all (chunk, window) tiles of the sort
    msm_plan pm = pl;
    if (fixed) {
        pm.windows = 1;
        pm.chunks = pl.windows * pl.chunks;
        pm.total_buckets = pl.buckets_per_window;
    }
    if (workspace_bytes < msm_workspace_bytes<F>(n) || !d_workspace) return ZKLC_ERR_INVALID_ARG;
    u32 seg_per_window = (pl.buckets_per_window + MSM_SEG - 1) / MSM_SEG;
    // carve the workspace (256-byte aligned pieces; sized for the plain form, which needs more)
    char *p = (char *)d_workspace;
    auto take = [&](size_t bytes) {
        char *r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    u32 *heavy_count = (u32 *)take(16);                 // zeroed every call
    u32 *totals = (u32 *)take((size_t)pl.total_buckets * 4);
    u32 *offsets = (u32 *)take((size_t)pl.total_buckets * 4);
    u32 nblocks = (pl.total_buckets + SCAN_ITEMS - 1) / SCAN_ITEMS;
    u32 *block_sums = (u32 *)take((size_t)(nblocks + 1) * 4);
    u32 *tile_cnt = (u32 *)take((size_t)pl.total_buckets * pl.chunks * 4);
    unsigned short *dig = (unsigned short *)take((size_t)pl.n_pad * pl.windows * 2);
    u32 *heavy_list = (u32 *)take(MSM_MAX_HEAVY * 4);
    u32 *entries = (u32 *)take((size_t)pl.n * pl.windows * 4 + 4);
    i32 *buckets = (i32 *)take((size_t)pl.total_buckets * XB);
    i32 *cpoints = (i32 *)take((size_t)pl.n * 2 * F::LIMBS * 4);
    const u32 slices = (u32)(((uint64_t)pl.n * pl.windows + MSM_SLICE - 1) / MSM_SLICE);
    i32 *partials = (i32 *)take(((size_t)slices + 1) * 2 * XB);
    i32 *seg_out = (i32 *)take((size_t)seg_per_window * pl.windows * XB);
    i32 *win_out = (i32 *)take((size_t)pl.windows * XB);
    const int BLK = msm_cfg<F>::BLOCK;
    const size_t lds = (size_t)pl.buckets_per_window * 4;

    ZKLC_HIP(ctx, hipMemsetAsync(heavy_count, 0, 16, st));
    const char *pkv = getenv("ZKLC_MSM_PACKED");          // A/B: 0 = point records of ten 32-bit limbs per coordinate (default packed)
    const bool packed = fixed || !(pkv && pkv[0] == '0');          // a table holds packed records
    if (pl.n && fixed) {
        hipLaunchKernelGGL((msm_recode_fixed_kernel<F>), dim3((pl.n_pad + 255) / 256), dim3(256), 0, st, fixed_table, d_scalars, pl, dig);
        hipLaunchKernelGGL(msm_hist_kernel, dim3(pl.chunks, pl.windows), dim3(MSM_SORT_THREADS), lds, st, (const unsigned short *)dig, pl,
                           tile_cnt);
    } else if (pl.n) {
        if constexpr (F::LIMBS == 10) {
            if (pl.glv) {
                const unsigned grid = (pl.n_pts + 255) / 256;
                if (packed)
                    hipLaunchKernelGGL(msm_glv_prepare_kernel<true>, dim3(grid), dim3(256), 0, st, d_points, d_scalars, pl, dig, cpoints);
                else
                    hipLaunchKernelGGL(msm_glv_prepare_kernel<false>, dim3(grid), dim3(256), 0, st, d_points, d_scalars, pl, dig, cpoints);
            }
        }
        if (!pl.glv)
            hipLaunchKernelGGL((msm_recode_kernel<AFF>), dim3((pl.n_pad + 255) / 256), dim3(256), 0, st, d_points, d_scalars, pl, dig);
        hipLaunchKernelGGL(msm_hist_kernel, dim3(pl.chunks, pl.windows), dim3(MSM_SORT_THREADS), lds, st, (const unsigned short *)dig, pl,
                           tile_cnt);
    } else {
        ZKLC_HIP(ctx, hipMemsetAsync(tile_cnt, 0, (size_t)pl.total_buckets * pl.chunks * 4, st));
    }
    const u32 nblocks_m = (pm.total_buckets + SCAN_ITEMS - 1) / SCAN_ITEMS;
    hipLaunchKernelGGL(msm_totals_kernel, dim3((pm.total_buckets + 255) / 256), dim3(256), 0, st, tile_cnt, pm, totals);
    hipLaunchKernelGGL(msm_scan_block_kernel, dim3(nblocks_m), dim3(256), 0, st, (const u32 *)totals, offsets, block_sums, pm.total_buckets);
    hipLaunchKernelGGL(msm_scan_sums_kernel, dim3(1), dim3(256), 0, st, block_sums, nblocks_m);
    hipLaunchKernelGGL(msm_scan_add_kernel, dim3((pm.total_buckets + 255) / 256), dim3(256), 0, st, offsets, (const u32 *)block_sums,
                       pm.total_buckets);
    if (pl.n) {
        hipLaunchKernelGGL(msm_scatter_kernel, dim3(pl.chunks, pl.windows), dim3(MSM_SORT_THREADS), lds, st, (const unsigned short *)dig, pl,
                           (const u32 *)tile_cnt, (const u32 *)offsets, entries, fixed ? pl.n : 0u);
    }
    // the bucket array starts as infinity (all-zero limbs): empty buckets are never written
    ZKLC_HIP(ctx, hipMemsetAsync(buckets, 0, (size_t)pm.total_buckets * XB, st));
    const i32 *recs = fixed ? fixed_table : (const i32 *)cpoints;
    if (pl.n) {
        // A/B switches: ZKLC_MSM_PACKED = 0 / 1 (point records of 10 x 32-bit limbs per coordinate / packed 8 x 32 bits, default packed),
        // ZKLC_MSM_WAVES = 1..3: waves per SIMD the G1 slice kernel's register allocation keeps
        if (!pl.glv && !fixed) {
            if (packed)
                hipLaunchKernelGGL((msm_convert_kernel<F, true>), dim3((pl.n + 255) / 256), dim3(256), 0, st, d_points, pl.n, cpoints);
            else
                hipLaunchKernelGGL((msm_convert_kernel<F, false>), dim3((pl.n + 255) / 256), dim3(256), 0, st, d_points, pl.n, cpoints);
        }
        const char *v = getenv("ZKLC_MSM_WAVES");
        int waves = (v && v[0] >= '1' && v[0] <= '3') ? v[0] - '0' : MSM_SLICE_WAVES_G1;
        if (F::LIMBS != 10) waves = 1;       // the Fp2 kernel needs the whole register file
        dim3 g((slices + MSM_SLICE_BLOCK - 1) / MSM_SLICE_BLOCK), b(MSM_SLICE_BLOCK);
#define MSM_SLICE_LAUNCH(W)                                                                                                          \
    do {                                                                                                                             \
        if (packed)                                                                                                                  \
            hipLaunchKernelGGL((msm_slice_kernel<F, W, true>), g, b, 0, st, recs, (const u32 *)entries,                              \
                               (const u32 *)offsets, (const u32 *)totals, pm, buckets, partials);                                   \
        else                                                                                                                         \
            hipLaunchKernelGGL((msm_slice_kernel<F, W, false>), g, b, 0, st, recs, (const u32 *)entries,                             \
                               (const u32 *)offsets, (const u32 *)totals, pm, buckets, partials);                                   \
    } while (0)
        if constexpr (F::LIMBS != 10) {
            MSM_SLICE_LAUNCH(1);
        } else {
            switch (waves) {
                case 1: MSM_SLICE_LAUNCH(1); break;
                case 3: MSM_SLICE_LAUNCH(3); break;
                default: MSM_SLICE_LAUNCH(2); break;
            }
        }
#undef MSM_SLICE_LAUNCH
    }
    hipLaunchKernelGGL(msm_combine_kernel<F>, dim3((pm.total_buckets + 63) / 64), dim3(64), 0, st, (const u32 *)offsets, (const u32 *)totals,
                       pm, (const i32 *)partials, buckets, heavy_list, heavy_count);
    hipLaunchKernelGGL(msm_heavy_combine_kernel<F>, dim3(MSM_MAX_HEAVY), dim3(BLK), 0, st, (const u32 *)offsets, (const u32 *)totals,
                       (const i32 *)partials, buckets, (const u32 *)heavy_list, (const u32 *)heavy_count);
    hipLaunchKernelGGL(msm_segment_kernel<F>, dim3((seg_per_window * pm.windows + 63) / 64), dim3(64), 0, st, (const i32 *)buckets, pm,
                       seg_out);
    hipLaunchKernelGGL(msm_window_kernel<F>, dim3(pm.windows), dim3(BLK), 0, st, (const i32 *)seg_out, pm, win_out);
    hipLaunchKernelGGL(msm_final_kernel<F>, dim3(1), dim3(BLK), 0, st, (const i32 *)win_out, pm, d_out_affine, d_out_inf);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

template <class F>
static int32_t msm_run_host(zklc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, uint64_t n, uint64_t *out_affine,
                            uint32_t *out_is_infinity) {
    const size_t PB = msm_cfg<F>::AFF * 8;
    if (!ctx || !out_affine || !out_is_infinity || (n && (!points || !scalars)) || n >= (1ULL << 31)) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    void *dp, *ds, *dw, *dout;
    int32_t rc;
    uint64_t wb = msm_workspace_bytes<F>(n);
    if ((rc = zklc_stage(ctx, 0, n ? n * PB : PB, &dp))) return rc;
    if ((rc = zklc_stage(ctx, 1, n ? n * 32 : 32, &ds))) return rc;
    if ((rc = zklc_stage(ctx, 2, wb, &dw))) return rc;
    if ((rc = zklc_stage(ctx, 3, 256, &dout))) return rc;
    if (n) {
        ZKLC_HIP(ctx, hipMemcpyAsync(dp, points, n * PB, hipMemcpyHostToDevice, ctx->stream));
        ZKLC_HIP(ctx, hipMemcpyAsync(ds, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = msm_run_dev<F>(ctx, ctx->stream, (const uint64_t *)dp, (const uint64_t *)ds, n, (uint64_t *)dout, (uint32_t *)((char *)dout + PB),
                        dw, wb);
    if (rc) return rc;
    ZKLC_HIP(ctx, zklc_readback_async(out_affine, dout, PB, ctx->stream));
    ZKLC_HIP(ctx, zklc_readback_async(out_is_infinity, (char *)dout + PB, 4, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}

extern "C" uint64_t zklc_bn254_g1_msm_workspace_bytes(uint64_t n) { return msm_workspace_bytes<FpField>(n); }
extern "C" int32_t zklc_bn254_g1_msm_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, const uint64_t *d_scalars, uint64_t n,
                                         uint64_t *d_out_affine, uint32_t *d_out_inf, void *d_workspace, uint64_t workspace_bytes) {
    return msm_run_dev<FpField>(ctx, stream, d_points, d_scalars, n, d_out_affine, d_out_inf, d_workspace, workspace_bytes);
}
extern "C" int32_t zklc_bn254_g1_msm(zklc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, uint64_t n, uint64_t *out_affine,
                                     uint32_t *out_is_infinity) {
    return msm_run_host<FpField>(ctx, points, scalars, n, out_affine, out_is_infinity);
}
// ---- fixed-base form: table construction and the multi-exponentiation over a table
template <class F>
static uint64_t msm_fixed_table_bytes(uint64_t n) {
    msm_plan pl = msm_make_plan(n, false);
    return (uint64_t)MSM_TABLE_HEADER_WORDS * 4 + (uint64_t)pl.windows * pl.n * msm_rec<F, true>::WORDS * 4;
}
template <class F>
static int32_t msm_fixed_table_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, uint64_t n, void *d_table, uint64_t table_bytes) {
    if (!ctx || !d_table || (n && !d_points) || n >= (1ULL << 30) || ((uintptr_t)d_table & 255)) return ZKLC_ERR_INVALID_ARG;
    if (table_bytes < msm_fixed_table_bytes<F>(n)) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = zklc_pick_stream(ctx, stream);
    msm_plan pl = msm_make_plan(n, false);
    u32 hdr[MSM_TABLE_HEADER_WORDS] = {MSM_TABLE_MAGIC, pl.n, pl.c, pl.windows, (u32)msm_rec<F, true>::WORDS};
    ZKLC_HIP(ctx, hipMemcpyAsync(d_table, hdr, sizeof(hdr), hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, zklc_stream_wait(st));                 // hdr lives on this stack frame
    {
        std::lock_guard<std::mutex> lk(ctx->msm_tables_mu);
        ctx->msm_tables[d_table] = {hdr[0], hdr[1], hdr[2], hdr[3], hdr[4]};
    }
    if (pl.n)
        hipLaunchKernelGGL((msm_fixed_table_kernel<F>), dim3((pl.n + 63) / 64), dim3(64), 0, st, d_points, pl.n, pl.c, pl.windows,
                           (i32 *)d_table + MSM_TABLE_HEADER_WORDS);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}
template <class F>
static int32_t msm_fixed_run_dev(zklc_ctx *ctx, void *stream, const void *d_table, const uint64_t *d_scalars, uint64_t n,
                                 uint64_t *d_out_affine, uint32_t *d_out_inf, void *d_workspace, uint64_t workspace_bytes) {
    if (!ctx || !d_table || ((uintptr_t)d_table & 255)) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    // is this a table for n bases of this group?  A table this context built (or has checked before) is known by its address: the
    // call stays enqueue-only.  Any other address is read ONCE -- 20 bytes on the caller's stream, a blocking wait -- and remembered.
    std::array<u32, 5> hdr;
    bool known = false;
    {
        std::lock_guard<std::mutex> lk(ctx->msm_tables_mu);
        auto it = ctx->msm_tables.find(d_table);
        if (it != ctx->msm_tables.end()) {
            hdr = it->second;
            known = true;
        }
    }
    if (!known) {
        hipStream_t st = zklc_pick_stream(ctx, stream);
        u32 *h_hdr = nullptr;
        ZKLC_HIP(ctx, hipHostMalloc((void **)&h_hdr, sizeof(u32) * 5, hipHostMallocDefault));
        hipError_t e = hipMemcpyAsync(h_hdr, d_table, sizeof(u32) * 5, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = zklc_stream_wait(st);
        for (int i = 0; i < 5; i++) hdr[i] = h_hdr[i];
        (void)hipHostFree(h_hdr);
        ZKLC_HIP(ctx, e);
        if (hdr[0] == MSM_TABLE_MAGIC) {
            std::lock_guard<std::mutex> lk(ctx->msm_tables_mu);
            ctx->msm_tables[d_table] = hdr;
        }
    }
    msm_plan pl = msm_make_plan(n, false);
    if (hdr[0] != MSM_TABLE_MAGIC || hdr[1] != pl.n || hdr[2] != pl.c || hdr[3] != pl.windows || hdr[4] != (u32)msm_rec<F, true>::WORDS)
        return ZKLC_ERR_INVALID_ARG;
    return msm_run_dev<F>(ctx, stream, nullptr, d_scalars, n, d_out_affine, d_out_inf, d_workspace, workspace_bytes,
                          (const i32 *)d_table + MSM_TABLE_HEADER_WORDS);
}
extern "C" uint64_t zklc_bn254_g1_msm_fixed_table_bytes(uint64_t n) { return msm_fixed_table_bytes<FpField>(n); }
extern "C" int32_t zklc_bn254_g1_msm_fixed_table_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, uint64_t n, void *d_table,
                                                     uint64_t table_bytes) {
    return msm_fixed_table_dev<FpField>(ctx, stream, d_points, n, d_table, table_bytes);
}
extern "C" int32_t zklc_bn254_g1_msm_fixed_dev(zklc_ctx *ctx, void *stream, const void *d_table, const uint64_t *d_scalars, uint64_t n,
                                               uint64_t *d_out_affine, uint32_t *d_out_inf, void *d_workspace, uint64_t workspace_bytes) {
    return msm_fixed_run_dev<FpField>(ctx, stream, d_table, d_scalars, n, d_out_affine, d_out_inf, d_workspace, workspace_bytes);
}
extern "C" uint64_t zklc_bn254_g2_msm_fixed_table_bytes(uint64_t n) { return msm_fixed_table_bytes<Fp2Field>(n); }
extern "C" int32_t zklc_bn254_g2_msm_fixed_table_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, uint64_t n, void *d_table,
                                                     uint64_t table_bytes) {
    return msm_fixed_table_dev<Fp2Field>(ctx, stream, d_points, n, d_table, table_bytes);
}
extern "C" int32_t zklc_bn254_g2_msm_fixed_dev(zklc_ctx *ctx, void *stream, const void *d_table, const uint64_t *d_scalars, uint64_t n,
                                               uint64_t *d_out_affine, uint32_t *d_out_inf, void *d_workspace, uint64_t workspace_bytes) {
    return msm_fixed_run_dev<Fp2Field>(ctx, stream, d_table, d_scalars, n, d_out_affine, d_out_inf, d_workspace, workspace_bytes);
}

extern "C" uint64_t zklc_bn254_g2_msm_workspace_bytes(uint64_t n) { return msm_workspace_bytes<Fp2Field>(n); }
extern "C" int32_t zklc_bn254_g2_msm_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, const uint64_t *d_scalars, uint64_t n,
                                         uint64_t *d_out_affine, uint32_t *d_out_inf, void *d_workspace, uint64_t workspace_bytes) {
    return msm_run_dev<Fp2Field>(ctx, stream, d_points, d_scalars, n, d_out_affine, d_out_inf, d_workspace, workspace_bytes);
}
extern "C" int32_t zklc_bn254_g2_msm(zklc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, uint64_t n, uint64_t *out_affine,
                                     uint32_t *out_is_infinity) {
    return msm_run_host<Fp2Field>(ctx, points, scalars, n, out_affine, out_is_infinity);
}
