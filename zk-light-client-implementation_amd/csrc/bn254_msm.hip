// BN254 G1 and G2 multi-scalar multiplication (Pippenger bucket method) for gfx950 + C ABI.
//
// Replaces gnark-crypto's `bn254.G1Affine.MultiExp` / `bn254.G2Affine.MultiExp` (un-vendored,
// gnark-plonky2-verifier/go.mod:9) under `groth16.Prove`
// (gnark-plonky2-verifier/cmd/web-api.go:77, tests/prover_test.go:64).  Every kernel is a template over the
// coordinate field (Fp for G1, Fp2 for G2; bn254_ec.cuh) -- the pipeline is identical.
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
#include "bn254_g2.cuh"
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
    // the window recoding covers 254 bits: a scalar that is not reduced (>= r, anything up to 2^256 - 1) is reduced here -- the
    // points have order r, so the sum is the same -- instead of silently losing its top bits.  Reduced scalars leave at the
    // first comparison of the top word.
    const u32 R[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
#pragma unroll 1
    for (int it = 0; it < 6; it++) {       // 2^256 / r < 6
        bool ge = true;
        for (int k = 7; k >= 0; k--)
            if (w[k] != R[k]) {
                ge = w[k] > R[k];
                break;
            }
        if (!ge) break;
        u64 borrow = 0;
        for (int k = 0; k < 8; k++) {
            u64 d = (u64)w[k] - R[k] - borrow;
            w[k] = (u32)d;
            borrow = (d >> 32) & 1;
        }
    }
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

template <int AFF>  // u64 words per affine point: 8 (G1) or 16 (G2)
ZKLC_D bool msm_point_is_inf(const u64 *points, u32 i) {
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(points + (size_t)i * AFF);
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < AFF / 2; k++) {
        ulonglong2 a = p[k];
        acc |= a.x | a.y;
    }
    return acc == 0;  // gnark encodes infinity as all-zero coordinates
}

template <bool SCATTER, int AFF>
__global__ void __launch_bounds__(256)
msm_digits_kernel(const u64 *__restrict__ points, const u64 *__restrict__ scalars, msm_plan pl, u32 *__restrict__ counts,
                  const u32 *__restrict__ offsets, u32 *__restrict__ cursor, u32 *__restrict__ entries) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pl.n) return;
    if (msm_point_is_inf<AFF>(points, i)) return;
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

// ---- bucket accumulation (template over the coordinate field F: FpField = G1, Fp2Field = G2)
template <class F>
struct msm_cfg {
    static constexpr int XYZZ = 4 * F::LIMBS;            // i32 words of a stored XYZZ point
    static constexpr int AFF = 2 * 4 * F::LIMBS / 10;    // u64 words of an affine point at the ABI
    static constexpr int BLOCK = F::LIMBS == 10 ? 256 : 128;  // workgroup size of the LDS tree reductions (<= 40 KiB of LDS)
};

template <class F>
ZKLC_D void msm_load_point(const u64 *points, u32 idx, typename F::T &x, typename F::T &y) {
    const int W = msm_cfg<F>::AFF;  // u32 words per coordinate
    const uint4 *p = reinterpret_cast<const uint4 *>(points + (size_t)idx * W);
    u32 w[2 * W];
#pragma unroll
    for (int k = 0; k < W / 2; k++) {
        uint4 a = p[k];
        w[4 * k] = a.x;
        w[4 * k + 1] = a.y;
        w[4 * k + 2] = a.z;
        w[4 * k + 3] = a.w;
    }
    x = F::from_gnark(w);
    y = F::from_gnark(w + W);
}

template <class F>
ZKLC_D void msm_store_xyzz(i32 *dst, const ec_xyzz<F> &p) {
    F::store(dst, p.X);
    F::store(dst + F::LIMBS, p.Y);
    F::store(dst + 2 * F::LIMBS, p.ZZ);
    F::store(dst + 3 * F::LIMBS, p.ZZZ);
}
template <class F>
ZKLC_D ec_xyzz<F> msm_load_xyzz(const i32 *src) {
    ec_xyzz<F> p;
    p.X = F::load(src);
    p.Y = F::load(src + F::LIMBS);
    p.ZZ = F::load(src + 2 * F::LIMBS);
    p.ZZZ = F::load(src + 3 * F::LIMBS);
    return p;
}

template <class F>
__global__ void __launch_bounds__(64)
msm_bucket_sum_kernel(const u64 *__restrict__ points, const u32 *__restrict__ entries, const u32 *__restrict__ offsets,
                      const u32 *__restrict__ counts, msm_plan pl, i32 *__restrict__ buckets, u32 *__restrict__ heavy_list,
                      u32 *__restrict__ heavy_count) {
    u32 key = blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= pl.total_buckets) return;
    u32 cnt = counts[key];
    ec_xyzz<F> acc = ec_infinity<F>();
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
        typename F::T x, y;
        msm_load_point<F>(points, ent >> 1, x, y);
        acc = ec_add_affine<F>(acc, x, y, ent & 1);
    }
    msm_store_xyzz<F>(buckets + (size_t)key * msm_cfg<F>::XYZZ, acc);
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
msm_heavy_bucket_kernel(const u64 *__restrict__ points, const u32 *__restrict__ entries, const u32 *__restrict__ offsets,
                        const u32 *__restrict__ counts, i32 *__restrict__ buckets, const u32 *__restrict__ heavy_list,
                        const u32 *__restrict__ heavy_count) {
    __shared__ i32 lds[msm_cfg<F>::BLOCK * msm_cfg<F>::XYZZ];
    u32 nheavy = *heavy_count;
    if (nheavy > MSM_MAX_HEAVY) nheavy = MSM_MAX_HEAVY;
    if (blockIdx.x >= nheavy) return;
    u32 key = heavy_list[blockIdx.x];
    u32 beg = offsets[key], cnt = counts[key];
    ec_xyzz<F> acc = ec_infinity<F>();
    for (u32 e = threadIdx.x; e < cnt; e += msm_cfg<F>::BLOCK) {
        u32 ent = entries[beg + e];
        typename F::T x, y;
        msm_load_point<F>(points, ent >> 1, x, y);
        acc = ec_add_affine<F>(acc, x, y, ent & 1);
    }
    acc = msm_block_reduce<F>(acc, lds);
    if (threadIdx.x == 0) msm_store_xyzz<F>(buckets + (size_t)key * msm_cfg<F>::XYZZ, acc);
}

// k * p for a small k (k < 2^31), double-and-add
template <class F>
ZKLC_D ec_xyzz<F> msm_small_mul(const ec_xyzz<F> &p, u32 k) {
    ec_xyzz<F> r = ec_infinity<F>();
    if (k == 0) return r;
    for (int b = 31 - __clz(k); b >= 0; b--) {
        r = ec_double(r);
        if ((k >> b) & 1) r = ec_add(r, p);
    }
    return r;
}

// one lane per segment of MSM_SEG buckets: sum_{b in seg} (b + 1) B_b  (b = bucket index within the window)
template <class F>
__global__ void __launch_bounds__(64) msm_segment_kernel(const i32 *__restrict__ buckets, msm_plan pl, i32 *__restrict__ seg_out) {
    const int XY = msm_cfg<F>::XYZZ;
    u32 seg_per_window = (pl.buckets_per_window + MSM_SEG - 1) / MSM_SEG;
    u32 s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= seg_per_window * pl.windows) return;
    u32 w = s / seg_per_window, si = s % seg_per_window;
    u32 lo = si * MSM_SEG, hi = lo + MSM_SEG < pl.buckets_per_window ? lo + MSM_SEG : pl.buckets_per_window;
    ec_xyzz<F> S = ec_infinity<F>(), T = ec_infinity<F>();
    for (u32 b = hi; b-- > lo;) {
        S = ec_add(S, msm_load_xyzz<F>(buckets + ((size_t)w * pl.buckets_per_window + b) * XY));
        T = ec_add(T, S);
    }
    // T = sum (b - lo + 1) B_b ; add lo * S
    if (lo) T = ec_add(T, msm_small_mul<F>(S, lo));
    msm_store_xyzz<F>(seg_out + (size_t)s * XY, T);
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

// result = sum_w 2^(c w) W_w ; affine output in gnark Montgomery words + infinity flag
template <class F>
__global__ void __launch_bounds__(msm_cfg<F>::BLOCK) msm_final_kernel(const i32 *__restrict__ win_out, msm_plan pl, u64 *__restrict__ out_affine, u32 *__restrict__ out_inf) {
    const int XY = msm_cfg<F>::XYZZ;
    __shared__ i32 lds[msm_cfg<F>::BLOCK * msm_cfg<F>::XYZZ];
    ec_xyzz<F> acc = ec_infinity<F>();
    if (threadIdx.x < pl.windows) {
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

// ---------------------------------------------------------------- host
static u32 msm_pick_window(u64 n) {
    if (n >= (1u << 19)) return 16;
    if (n >= (1u << 15)) return 14;
    if (n >= (1u << 11)) return 11;
    if (n >= 64) return 8;
    return 4;
}

template <class F>
static uint64_t msm_workspace_bytes(uint64_t n) {
    const uint64_t XB = msm_cfg<F>::XYZZ * 4;
    u32 c = msm_pick_window(n), windows = 254 / c + 1, bpw = 1u << (c - 1), total = windows * bpw;
    u32 seg_per_window = (bpw + MSM_SEG - 1) / MSM_SEG;
    uint64_t b = 0;
    b += (uint64_t)total * 4 * 3;                       // counts, offsets, cursor
    b += ((uint64_t)total / SCAN_ITEMS + 2) * 4;        // scan block sums
    b += n * windows * 4;                               // entries
    b += (uint64_t)total * XB;                          // buckets
    b += (uint64_t)seg_per_window * windows * XB;       // segment sums
    b += (uint64_t)windows * XB;                        // window sums
    b += (MSM_MAX_HEAVY + 4) * 4;                       // heavy list + counter
    b += 256 * 16;                                      // alignment slack
    return b;
}

template <class F>
static int32_t msm_run_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, const uint64_t *d_scalars, uint64_t n,
                           uint64_t *d_out_affine, uint32_t *d_out_inf, void *d_workspace, uint64_t workspace_bytes) {
    const int AFF = msm_cfg<F>::AFF;
    const size_t XB = msm_cfg<F>::XYZZ * 4;
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
    if (workspace_bytes < msm_workspace_bytes<F>(n) || !d_workspace) return ZKLC_ERR_INVALID_ARG;
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
    i32 *buckets = (i32 *)take((size_t)pl.total_buckets * XB);
    i32 *seg_out = (i32 *)take((size_t)seg_per_window * pl.windows * XB);
    i32 *win_out = (i32 *)take((size_t)pl.windows * XB);
    const int BLK = msm_cfg<F>::BLOCK;

    ZKLC_HIP(ctx, hipMemsetAsync(counts, 0, zero_bytes, st));
    u32 gpts = (pl.n + 255) / 256;
    if (pl.n) {
        hipLaunchKernelGGL((msm_digits_kernel<false, AFF>), dim3(gpts), dim3(256), 0, st, d_points, d_scalars, pl, counts,
                           (const u32 *)nullptr, cursor, entries);
    }
    hipLaunchKernelGGL(msm_scan_block_kernel, dim3(nblocks), dim3(256), 0, st, (const u32 *)counts, offsets, block_sums, pl.total_buckets);
    hipLaunchKernelGGL(msm_scan_sums_kernel, dim3(1), dim3(256), 0, st, block_sums, nblocks);
    hipLaunchKernelGGL(msm_scan_add_kernel, dim3((pl.total_buckets + 255) / 256), dim3(256), 0, st, offsets, (const u32 *)block_sums,
                       pl.total_buckets);
    if (pl.n) {
        hipLaunchKernelGGL((msm_digits_kernel<true, AFF>), dim3(gpts), dim3(256), 0, st, d_points, d_scalars, pl, counts,
                           (const u32 *)offsets, cursor, entries);
    }
    hipLaunchKernelGGL(msm_bucket_sum_kernel<F>, dim3((pl.total_buckets + 63) / 64), dim3(64), 0, st, d_points, (const u32 *)entries,
                       (const u32 *)offsets, (const u32 *)counts, pl, buckets, heavy_list, heavy_count);
    hipLaunchKernelGGL(msm_heavy_bucket_kernel<F>, dim3(MSM_MAX_HEAVY), dim3(BLK), 0, st, d_points, (const u32 *)entries,
                       (const u32 *)offsets, (const u32 *)counts, buckets, (const u32 *)heavy_list, (const u32 *)heavy_count);
    hipLaunchKernelGGL(msm_segment_kernel<F>, dim3((seg_per_window * pl.windows + 63) / 64), dim3(64), 0, st, (const i32 *)buckets, pl,
                       seg_out);
    hipLaunchKernelGGL(msm_window_kernel<F>, dim3(pl.windows), dim3(BLK), 0, st, (const i32 *)seg_out, pl, win_out);
    hipLaunchKernelGGL(msm_final_kernel<F>, dim3(1), dim3(BLK), 0, st, (const i32 *)win_out, pl, d_out_affine, d_out_inf);
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
    ZKLC_HIP(ctx, hipMemcpyAsync(out_affine, dout, PB, hipMemcpyDeviceToHost, ctx->stream));
    ZKLC_HIP(ctx, hipMemcpyAsync(out_is_infinity, (char *)dout + PB, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZKLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
extern "C" uint64_t zklc_bn254_g2_msm_workspace_bytes(uint64_t n) { return msm_workspace_bytes<Fp2Field>(n); }
extern "C" int32_t zklc_bn254_g2_msm_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, const uint64_t *d_scalars, uint64_t n,
                                         uint64_t *d_out_affine, uint32_t *d_out_inf, void *d_workspace, uint64_t workspace_bytes) {
    return msm_run_dev<Fp2Field>(ctx, stream, d_points, d_scalars, n, d_out_affine, d_out_inf, d_workspace, workspace_bytes);
}
extern "C" int32_t zklc_bn254_g2_msm(zklc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, uint64_t n, uint64_t *out_affine,
                                     uint32_t *out_is_infinity) {
    return msm_run_host<Fp2Field>(ctx, points, scalars, n, out_affine, out_is_infinity);
}
