// BN254 G1 multi-scalar multiplication (Pippenger bucket method) for gfx950 + C ABI.
//
// Replaces gnark-crypto's `bn254.G1Affine.MultiExp` (un-vendored,
// gnark-plonky2-verifier/go.mod:9) under `groth16.Prove`
// (gnark-plonky2-verifier/cmd/web-api.go:77, tests/prover_test.go:64).
//
// Pipeline (all on the device, no host round trips):
//   1. signed c-bit digits of every scalar; histogram of (window, |digit|) keys
//   2. exclusive scan of the histogram -> bucket offsets
//   3. counting-sort scatter of (point index, sign) entries into bucket order
//   4. bucket accumulation: one LANE per bucket walks its entries and adds the affine
//      points (extended-Jacobian mixed addition); buckets with more than
//      MSM_HEAVY entries (skewed scalars, e.g. many equal to 1) go to a
//      workgroup-per-bucket kernel with an LDS tree reduction instead
//   5. bucket reduction: per window, segments of 128 buckets -> running-sum
//      trick + small scalar multiple, then an LDS tree over the segments
//   6. 2^(c w) * window_w by doublings (one lane per window), tree sum, affine output
// Curve additions are order-independent as group elements, so the (arbitrary) order of the
// atomics in steps 1 and 3 never changes the affine result.
#include "bn254_g1.cuh"
#include "zklc_internal.h"

#define MSM_MAX_WINDOWS 32
#define MSM_HEAVY 4096u
#define MSM_MAX_HEAVY 2048u
#define MSM_SEG 128u

struct msm_plan {
    u32 n, c, windows, buckets_per_window, total_buckets;
};

ZKLC_D void msm_load_scalar(const u64 *scalars, u32 i, u32 *w) {
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(scalars + (size_t)i * 4);
    ulonglong2 a = p[0], b = p[1];
    w[0] = (u32)a.x; w[1] = (u32)(a.x >> 32); w[2] = (u32)a.y; w[3] = (u32)(a.y >> 32);
    w[4] = (u32)b.x; w[5] = (u32)(b.x >> 32); w[6] = (u32)b.y; w[7] = (u32)(b.y >> 32);
}

// signed digit of window w given the running carry (updated): digit in [-(2^(c-1) - 1), 2^(c-1)]
ZKLC_D int msm_digit(const u32 *sw, u32 w, u32 c, u32 &carry) {
    u32 bit = w * c, wi = bit >> 5, sh = bit & 31;
    u64 x = (u64)sw[wi] >> sh;
    if (wi + 1 < 8) x |= (u64)sw[wi + 1] << (32 - sh);
    u32 raw = ((u32)x & ((1u << c) - 1)) + carry;
    if (raw > (1u << (c - 1))) {
        carry = 1;
        return (int)raw - (int)(1u << c);
    }
    carry = 0;
    return (int)raw;
}

ZKLC_D bool msm_point_is_inf(const u64 *points, u32 i) {
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(points + (size_t)i * 8);
    ulonglong2 a = p[0], b = p[1], c = p[2], d = p[3];
    return (a.x | a.y | b.x | b.y | c.x | c.y | d.x | d.y) == 0;  // gnark encodes infinity as (0, 0)
}

template <bool SCATTER>
__global__ void __launch_bounds__(256)
msm_digits_kernel(const u64 *__restrict__ points, const u64 *__restrict__ scalars, msm_plan pl, u32 *__restrict__ counts,
                  const u32 *__restrict__ offsets, u32 *__restrict__ cursor, u32 *__restrict__ entries) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pl.n) return;
    if (msm_point_is_inf(points, i)) return;
    u32 sw[8];
    msm_load_scalar(scalars, i, sw);
    u32 carry = 0;
    for (u32 w = 0; w < pl.windows; w++) {
        int d = msm_digit(sw, w, pl.c, carry);
        if (d == 0) continue;
        u32 neg = d < 0;
        u32 key = w * pl.buckets_per_window + (u32)(neg ? -d : d) - 1;
        if (SCATTER) {
            u32 pos = atomicAdd(&cursor[key], 1u);
            entries[offsets[key] + pos] = (i << 1) | neg;
        } else {
            atomicAdd(&counts[key], 1u);
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

// ---- bucket accumulation
ZKLC_D void msm_load_point(const u64 *points, u32 idx, fp &x, fp &y) {
    const uint4 *p = reinterpret_cast<const uint4 *>(points + (size_t)idx * 8);
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    u32 wx[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32 wy[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    x = fp_from_gnark(wx);
    y = fp_from_gnark(wy);
}

ZKLC_D void msm_store_xyzz(i32 *dst, const g1_xyzz &p) {
#pragma unroll
    for (int k = 0; k < 10; k++) {
        dst[k] = p.X.v[k];
        dst[10 + k] = p.Y.v[k];
        dst[20 + k] = p.ZZ.v[k];
        dst[30 + k] = p.ZZZ.v[k];
    }
}
ZKLC_D g1_xyzz msm_load_xyzz(const i32 *src) {
    g1_xyzz p;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        p.X.v[k] = src[k];
        p.Y.v[k] = src[10 + k];
        p.ZZ.v[k] = src[20 + k];
        p.ZZZ.v[k] = src[30 + k];
    }
    return p;
}

__global__ void __launch_bounds__(64)
msm_bucket_sum_kernel(const u64 *__restrict__ points, const u32 *__restrict__ entries, const u32 *__restrict__ offsets,
                      const u32 *__restrict__ counts, msm_plan pl, i32 *__restrict__ buckets, u32 *__restrict__ heavy_list,
                      u32 *__restrict__ heavy_count) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= pl.total_buckets) return;
    u32 cnt = counts[key];
    g1_xyzz acc = g1_infinity();
    if (cnt > MSM_HEAVY) {
        u32 slot = atomicAdd(heavy_count, 1u);
        if (slot < MSM_MAX_HEAVY) {
            heavy_list[slot] = key;
            return;  // the heavy kernel writes this bucket
        }
        // list full: fall through and do it here (slow but correct)
    }
    u32 beg = offsets[key];
    for (u32 e = 0; e < cnt; e++) {
        u32 ent = entries[beg + e];
        fp x, y;
        msm_load_point(points, ent >> 1, x, y);
        acc = g1_add_affine(acc, x, y, ent & 1);
    }
    msm_store_xyzz(buckets + (size_t)key * 40, acc);
}

// LDS tree reduction of one XYZZ point per thread (256 threads); result in thread 0
ZKLC_D g1_xyzz msm_block_reduce(g1_xyzz acc, i32 *lds /* 256 * 40 words */) {
    for (u32 stride = 128; stride >= 1; stride >>= 1) {
        if (threadIdx.x >= stride && threadIdx.x < 2 * stride) msm_store_xyzz(lds + (threadIdx.x - stride) * 40, acc);
        __syncthreads();
        if (threadIdx.x < stride) acc = g1_add(acc, msm_load_xyzz(lds + threadIdx.x * 40));
        __syncthreads();
    }
    return acc;
}

__global__ void __launch_bounds__(256)
msm_heavy_bucket_kernel(const u64 *__restrict__ points, const u32 *__restrict__ entries, const u32 *__restrict__ offsets,
                        const u32 *__restrict__ counts, i32 *__restrict__ buckets, const u32 *__restrict__ heavy_list,
                        const u32 *__restrict__ heavy_count) {
    __shared__ i32 lds[256 * 40];
    u32 nheavy = *heavy_count;
    if (nheavy > MSM_MAX_HEAVY) nheavy = MSM_MAX_HEAVY;
    if (blockIdx.x >= nheavy) return;
    u32 key = heavy_list[blockIdx.x];
    u32 beg = offsets[key], cnt = counts[key];
    g1_xyzz acc = g1_infinity();
    for (u32 e = threadIdx.x; e < cnt; e += 256) {
        u32 ent = entries[beg + e];
        fp x, y;
        msm_load_point(points, ent >> 1, x, y);
        acc = g1_add_affine(acc, x, y, ent & 1);
    }
    acc = msm_block_reduce(acc, lds);
    if (threadIdx.x == 0) msm_store_xyzz(buckets + (size_t)key * 40, acc);
}

// k * p for a small k (k < 2^31), double-and-add
ZKLC_D g1_xyzz msm_small_mul(const g1_xyzz &p, u32 k) {
    g1_xyzz r = g1_infinity();
    if (k == 0) return r;
    for (int b = 31 - __clz(k); b >= 0; b--) {
        r = g1_double(r);
        if ((k >> b) & 1) r = g1_add(r, p);
    }
    return r;
}

// one lane per segment of MSM_SEG buckets: sum_{b in seg} (b + 1) B_b  (b = bucket index within the window)
__global__ void __launch_bounds__(64) msm_segment_kernel(const i32 *__restrict__ buckets, msm_plan pl, i32 *__restrict__ seg_out) {
    u32 seg_per_window = (pl.buckets_per_window + MSM_SEG - 1) / MSM_SEG;
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= seg_per_window * pl.windows) return;
    u32 w = s / seg_per_window, si = s % seg_per_window;
    u32 lo = si * MSM_SEG, hi = lo + MSM_SEG < pl.buckets_per_window ? lo + MSM_SEG : pl.buckets_per_window;
    g1_xyzz S = g1_infinity(), T = g1_infinity();
    for (u32 b = hi; b-- > lo;) {
        S = g1_add(S, msm_load_xyzz(buckets + ((size_t)w * pl.buckets_per_window + b) * 40));
        T = g1_add(T, S);
    }
    // T = sum (b - lo + 1) B_b ; add lo * S
    if (lo) T = g1_add(T, msm_small_mul(S, lo));
    msm_store_xyzz(seg_out + (size_t)s * 40, T);
}

// one workgroup per window: sum of its segment results
__global__ void __launch_bounds__(256) msm_window_kernel(const i32 *__restrict__ seg_out, msm_plan pl, i32 *__restrict__ win_out) {
    __shared__ i32 lds[256 * 40];
    u32 seg_per_window = (pl.buckets_per_window + MSM_SEG - 1) / MSM_SEG;
    u32 w = blockIdx.x;
    g1_xyzz acc = g1_infinity();
    for (u32 s = threadIdx.x; s < seg_per_window; s += 256) acc = g1_add(acc, msm_load_xyzz(seg_out + ((size_t)w * seg_per_window + s) * 40));
    acc = msm_block_reduce(acc, lds);
    if (threadIdx.x == 0) msm_store_xyzz(win_out + (size_t)w * 40, acc);
}

// result = sum_w 2^(c w) W_w ; affine output in gnark Montgomery words (x: 4 u64, y: 4 u64) + infinity flag
__global__ void __launch_bounds__(256) msm_final_kernel(const i32 *__restrict__ win_out, msm_plan pl, u64 *__restrict__ out_affine, u32 *__restrict__ out_inf) {
    __shared__ i32 lds[256 * 40];
    g1_xyzz acc = g1_infinity();
    if (threadIdx.x < pl.windows) {
        acc = msm_load_xyzz(win_out + (size_t)threadIdx.x * 40);
        u32 dbl = pl.c * threadIdx.x;
        for (u32 k = 0; k < dbl; k++) acc = g1_double(acc);
    }
    acc = msm_block_reduce(acc, lds);
    if (threadIdx.x == 0) {
        u32 o[16];
        u32 inf = g1_to_affine_gnark(o, acc);
        for (int k = 0; k < 8; k++) out_affine[k] = (u64)o[2 * k] | ((u64)o[2 * k + 1] << 32);
        *out_inf = inf;
    }
}

// ---------------------------------------------------------------- host
static u32 msm_pick_window(u64 n) {
    if (n >= (1u << 19)) return 16;
    if (n >= (1u << 15)) return 14;
    if (n >= (1u << 11)) return 11;
    if (n >= 64) return 8;
    return 4;
}

extern "C" uint64_t zklc_bn254_g1_msm_workspace_bytes(uint64_t n) {
    u32 c = msm_pick_window(n), windows = 254 / c + 1, bpw = 1u << (c - 1), total = windows * bpw;
    u32 seg_per_window = (bpw + MSM_SEG - 1) / MSM_SEG;
    uint64_t b = 0;
    b += (uint64_t)total * 4 * 3;                       // counts, offsets, cursor
    b += ((uint64_t)total / SCAN_ITEMS + 2) * 4;        // scan block sums
    b += n * windows * 4;                               // entries
    b += (uint64_t)total * 160;                         // buckets
    b += (uint64_t)seg_per_window * windows * 160;      // segment sums
    b += (uint64_t)windows * 160;                       // window sums
    b += (MSM_MAX_HEAVY + 4) * 4;                       // heavy list + counter
    b += 256 * 16;                                      // alignment slack
    return b;
}

extern "C" int32_t zklc_bn254_g1_msm_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, const uint64_t *d_scalars, uint64_t n,
                                         uint64_t *d_out_affine, uint32_t *d_out_inf, void *d_workspace, uint64_t workspace_bytes) {
    if (!ctx || !d_out_affine || !d_out_inf || (n && (!d_points || !d_scalars)) || n >= (1ULL << 31)) return ZKLC_ERR_INVALID_ARG;
    if (((uintptr_t)d_points | (uintptr_t)d_scalars) & 15) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = zklc_pick_stream(ctx, stream);
    msm_plan pl;
    pl.n = (u32)n;
    pl.c = msm_pick_window(n);
    pl.windows = 254 / pl.c + 1;
    pl.buckets_per_window = 1u << (pl.c - 1);
    pl.total_buckets = pl.windows * pl.buckets_per_window;
    if (workspace_bytes < zklc_bn254_g1_msm_workspace_bytes(n) || !d_workspace) return ZKLC_ERR_INVALID_ARG;
    u32 seg_per_window = (pl.buckets_per_window + MSM_SEG - 1) / MSM_SEG;
    // carve the workspace (256-byte aligned pieces)
    char *p = (char *)d_workspace;
    auto take = [&](size_t bytes) {
        char *r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    u32 *counts = (u32 *)take((size_t)pl.total_buckets * 4);
    u32 *cursor = (u32 *)take((size_t)pl.total_buckets * 4);
    u32 *heavy_count = (u32 *)take(16);
    size_t zero_bytes = (char *)p - (char *)counts;  // counts, cursor, heavy counter are zeroed every call
    u32 *offsets = (u32 *)take((size_t)pl.total_buckets * 4);
    u32 nblocks = (pl.total_buckets + SCAN_ITEMS - 1) / SCAN_ITEMS;
    u32 *block_sums = (u32 *)take((size_t)(nblocks + 1) * 4);
    u32 *heavy_list = (u32 *)take(MSM_MAX_HEAVY * 4);
    u32 *entries = (u32 *)take((size_t)n * pl.windows * 4 + 4);
    i32 *buckets = (i32 *)take((size_t)pl.total_buckets * 160);
    i32 *seg_out = (i32 *)take((size_t)seg_per_window * pl.windows * 160);
    i32 *win_out = (i32 *)take((size_t)pl.windows * 160);

    ZKLC_HIP(ctx, hipMemsetAsync(counts, 0, zero_bytes, st));
    u32 gpts = (pl.n + 255) / 256;
    if (pl.n) {
        hipLaunchKernelGGL(msm_digits_kernel<false>, dim3(gpts), dim3(256), 0, st, d_points, d_scalars, pl, counts, (const u32 *)nullptr,
                           cursor, entries);
    }
    hipLaunchKernelGGL(msm_scan_block_kernel, dim3(nblocks), dim3(256), 0, st, (const u32 *)counts, offsets, block_sums, pl.total_buckets);
    hipLaunchKernelGGL(msm_scan_sums_kernel, dim3(1), dim3(256), 0, st, block_sums, nblocks);
    hipLaunchKernelGGL(msm_scan_add_kernel, dim3((pl.total_buckets + 255) / 256), dim3(256), 0, st, offsets, (const u32 *)block_sums,
                       pl.total_buckets);
    if (pl.n) {
        hipLaunchKernelGGL(msm_digits_kernel<true>, dim3(gpts), dim3(256), 0, st, d_points, d_scalars, pl, counts, (const u32 *)offsets,
                           cursor, entries);
    }
    hipLaunchKernelGGL(msm_bucket_sum_kernel, dim3((pl.total_buckets + 63) / 64), dim3(64), 0, st, d_points, (const u32 *)entries,
                       (const u32 *)offsets, (const u32 *)counts, pl, buckets, heavy_list, heavy_count);
    hipLaunchKernelGGL(msm_heavy_bucket_kernel, dim3(MSM_MAX_HEAVY), dim3(256), 0, st, d_points, (const u32 *)entries, (const u32 *)offsets,
                       (const u32 *)counts, buckets, (const u32 *)heavy_list, (const u32 *)heavy_count);
    hipLaunchKernelGGL(msm_segment_kernel, dim3((seg_per_window * pl.windows + 63) / 64), dim3(64), 0, st, (const i32 *)buckets, pl, seg_out);
    hipLaunchKernelGGL(msm_window_kernel, dim3(pl.windows), dim3(256), 0, st, (const i32 *)seg_out, pl, win_out);
    hipLaunchKernelGGL(msm_final_kernel, dim3(1), dim3(256), 0, st, (const i32 *)win_out, pl, d_out_affine, d_out_inf);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" int32_t zklc_bn254_g1_msm(zklc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, uint64_t n, uint64_t *out_affine,
                                     uint32_t *out_is_infinity) {
    if (!ctx || !out_affine || !out_is_infinity || (n && (!points || !scalars)) || n >= (1ULL << 31)) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    void *dp, *ds, *dw, *dout;
    int32_t rc;
    uint64_t wb = zklc_bn254_g1_msm_workspace_bytes(n);
    if ((rc = zklc_stage(ctx, 0, n ? n * 64 : 64, &dp))) return rc;
    if ((rc = zklc_stage(ctx, 1, n ? n * 32 : 32, &ds))) return rc;
    if ((rc = zklc_stage(ctx, 2, wb, &dw))) return rc;
    if ((rc = zklc_stage(ctx, 3, 128, &dout))) return rc;
    if (n) {
        ZKLC_HIP(ctx, hipMemcpyAsync(dp, points, n * 64, hipMemcpyHostToDevice, ctx->stream));
        ZKLC_HIP(ctx, hipMemcpyAsync(ds, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = zklc_bn254_g1_msm_dev(ctx, ctx->stream, (const uint64_t *)dp, (const uint64_t *)ds, n, (uint64_t *)dout,
                               (uint32_t *)((char *)dout + 64), dw, wb);
    if (rc) return rc;
    ZKLC_HIP(ctx, hipMemcpyAsync(out_affine, dout, 64, hipMemcpyDeviceToHost, ctx->stream));
    ZKLC_HIP(ctx, hipMemcpyAsync(out_is_infinity, (char *)dout + 64, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZKLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKLC_OK;
}
