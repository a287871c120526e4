// NTT over the BN254 scalar field Fr (2-adicity 28) for gfx950 + C ABI.
//
// Replaces gnark-crypto's `fft.Domain.FFT / FFTInverse` on ecc/bn254/fr (un-vendored, gnark-plonky2-verifier/go.mod:9),
// used by `groth16.Prove` (gnark-plonky2-verifier/cmd/web-api.go:77) to compute the quotient polynomial h from the
// A, B, C wire evaluations (three inverse FFTs, three coset FFTs, one coset inverse FFT of size = domain cardinality).
// Elements cross the ABI in gnark-crypto's memory layout (x * 2^256 mod r, 4 little-endian u64); inside, the ten-limb
// lazy Montgomery form of bn254_fr.cuh.  Definition: values[k] = sum_j coeffs[j] w^(jk), w = g28^(2^28 / n) with
// g28 = 0x2a3c09f0a58a7e8500e0a7eb8ef62abc402d111e41112ed49bd61b6e725b19f0 (gnark-crypto's rootOfUnity; order 2^28 is
// checked in oracle/bn254_fr.py); coset generator 5 (fr.Generator / FrMultiplicativeGen).
// Natural-order transforms of 2^12 .. 2^22 points (what `computeH` runs) take the two-pass form of bn254_fr_ntt_tile.cuh: two
// launches, each an LDS-resident sub-transform per workgroup, tables resident per context.  The other cases (smaller or larger
// sizes, bit-reversed input / output) keep round 1's one-launch-per-stage path below.
#include "bn254_fr_ntt_tile.cuh"
#include "zklc_internal.h"
#include <stdlib.h>

ZKLC_D fr frn_load_gnark(const u64 *p) {
    u32 w[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        w[2 * k] = (u32)p[k];
        w[2 * k + 1] = (u32)(p[k] >> 32);
    }
    return fr_reduce(fr_from_gnark(w));
}

// tw[i] = w^i, i < n/2, w = root28^(2^(28 - logn)) (or its inverse)
__global__ void frn_twiddle_kernel(i32 *tw, u32 logn, u32 inverse, u32 half) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const u32 rw[8] = FR_ROOT28_WORDS;
    fr g = fr_from_regular(rw);
    fr w = frn_pow(g, 1ULL << (28 - logn));
    if (inverse) w = fr_inv(w);
    frn_store(tw + (size_t)i * 10, frn_pow(w, i));
}

// consts[0] = 1/n, consts[1] = 5, consts[2] = 1/5 (internal Montgomery form), computed once per call by one lane
__global__ void frn_consts_kernel(i32 *consts, u32 logn) {
    if (threadIdx.x || blockIdx.x) return;
    const fr r2 = FR_R2;
    fr nn = fr_zero();
    nn.v[0] = (i32)((1u << logn) & 0x3ffffff);
    nn.v[1] = (i32)((1u << logn) >> 26);
    fr five = fr_zero();
    five.v[0] = 5;
    five = fr_mul(five, r2);
    frn_store(consts, fr_inv(fr_mul(nn, r2)));
    frn_store(consts + 10, five);
    frn_store(consts + 20, fr_inv(five));
}

// work[i] = data[src(i)] * shift^src(i)   (src = bit reversal when the input is in natural order)
__global__ void frn_load_kernel(const u64 *__restrict__ data, i32 *__restrict__ work, const i32 *__restrict__ consts, u32 logn,
                                u32 bitrev_in, u32 coset) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << logn)) return;
    u32 s = bitrev_in && logn ? (__brev(i) >> (32 - logn)) : i;
    fr a = frn_load_gnark(data + (size_t)s * 4);
    if (coset) a = fr_mul(a, frn_pow(frn_load(consts + 10), s));
    frn_store(work + (size_t)i * 10, a);
}

// one decimation-in-time stage: pairs (j, j + half) inside blocks of 2 * half
__global__ void __launch_bounds__(256) frn_stage_kernel(i32 *__restrict__ work, const i32 *__restrict__ tw, u32 logn, u32 stage) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (1u << (logn - 1))) return;
    u32 half = 1u << stage;
    u32 j = b & (half - 1);
    u32 i0 = ((b >> stage) << (stage + 1)) | j;
    u32 i1 = i0 + half;
    fr u = fr_reduce(frn_load(work + (size_t)i0 * 10));     // keeps the lazy magnitudes bounded across the stages
    fr v = fr_mul(frn_load(work + (size_t)i1 * 10), frn_load(tw + (size_t)(j << (logn - 1 - stage)) * 10));
    frn_store(work + (size_t)i0 * 10, fr_add(u, v));
    frn_store(work + (size_t)i1 * 10, fr_sub(u, v));
}

// data[dst(i)] = work[i] * scale * shift_inv^i
__global__ void frn_store_kernel(const i32 *__restrict__ work, u64 *__restrict__ data, const i32 *__restrict__ consts, u32 logn,
                                 u32 bitrev_out, u32 inverse, u32 coset) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << logn)) return;
    fr a = frn_load(work + (size_t)i * 10);
    if (inverse) {
        a = fr_mul(a, frn_load(consts));
        if (coset) a = fr_mul(a, frn_pow(frn_load(consts + 20), i));
    }
    u32 d = bitrev_out && logn ? (__brev(i) >> (32 - logn)) : i;
    u32 w[8];
    fr_to_gnark(w, a);
    u64 *o = data + (size_t)d * 4;
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = (u64)w[2 * k] | ((u64)w[2 * k + 1] << 32);
}

// a[i] = (a[i] * b[i] - c[i]) * scale: the quotient step of groth16 computeH between the coset FFTs and the coset inverse FFT
// (gnark backend/groth16/bn254/prove.go `computeH`: scale = 1 / (g^n - 1), the inverse of the vanishing polynomial on the coset)
__global__ void __launch_bounds__(256) frn_mul_sub_scale_kernel(u64 *__restrict__ a, const u64 *__restrict__ b, const u64 *__restrict__ c,
                                                                 const u64 s0, const u64 s1, const u64 s2, const u64 s3, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 sw[4] = {s0, s1, s2, s3};
    fr x = fr_mul(frn_load_gnark(a + i * 4), frn_load_gnark(b + i * 4));
    x = fr_mul(fr_sub(x, frn_load_gnark(c + i * 4)), frn_load_gnark(sw));
    u32 w[8];
    fr_to_gnark(w, x);
#pragma unroll
    for (int k = 0; k < 4; k++) a[i * 4 + k] = (u64)w[2 * k] | ((u64)w[2 * k + 1] << 32);
}

// ---------------------------------------------------------------- two-pass path (bn254_fr_ntt_tile.cuh)
__global__ void frn_table_consts_kernel(i32 *tab, u32 log_n, u32 inverse) {
    if (threadIdx.x == 0 && blockIdx.x == 0) frn_table_consts(tab, log_n, inverse);
}
__global__ void __launch_bounds__(256) frn_table_entries_kernel(i32 *tab, frn_plan p) {
    u32 e = 4 + blockIdx.x * blockDim.x + threadIdx.x;
    if (e < frn_table_elems(p)) frn_store(tab + (size_t)e * 10, frn_table_entry(tab, p, e));
}

#define FRN_TILE_THREADS 256
// the T stages of an Nt = 2^T point tile in LDS (a barrier per stage)
ZKLC_D void frn_tile_stages(i32 *tile, u32 T, const i32 *loc) {
    const u32 half_n = (1u << T) >> 1;
    for (u32 t = 0; t < T; t++) {
        __syncthreads();
        for (u32 b = threadIdx.x; b < half_n; b += FRN_TILE_THREADS) frn_tile_butterfly(tile, T, t, b, loc);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(FRN_TILE_THREADS)
frn_pass_a_kernel(const u64 *__restrict__ data, i32 *__restrict__ mid, const i32 *__restrict__ tab, frn_plan p, u32 coset_in) {
    extern __shared__ i32 frn_lds[];
    const u32 N1 = 1u << p.t1, j2 = blockIdx.x;
    for (u32 j1 = threadIdx.x; j1 < N1; j1 += FRN_TILE_THREADS)
        frn_tile_store(frn_lds, N1, frn_bitrev(j1, p.t1), frn_pass_a_in(data, tab, p, j1, j2, coset_in));
    frn_tile_stages(frn_lds, p.t1, tab + (size_t)frn_off_loc1(p) * 10);
    i32 *row = mid + (size_t)j2 * N1 * 10;                    // T[j2][k1], element-major (ten consecutive words per element)
    for (u32 k1 = threadIdx.x; k1 < N1; k1 += FRN_TILE_THREADS)
        frn_store(row + (size_t)k1 * 10, frn_pass_a_out(frn_tile_load(frn_lds, N1, k1), tab, p, k1, j2));
}

__global__ void __launch_bounds__(FRN_TILE_THREADS)
frn_pass_b_kernel(const i32 *__restrict__ mid, u64 *__restrict__ data, const i32 *__restrict__ tab, frn_plan p, u32 coset_out) {
    extern __shared__ i32 frn_lds[];
    const u32 N1 = 1u << p.t1, N2 = 1u << p.t2, k1 = blockIdx.x;
    for (u32 j2 = threadIdx.x; j2 < N2; j2 += FRN_TILE_THREADS)
        frn_tile_store(frn_lds, N2, frn_bitrev(j2, p.t2), frn_load(mid + ((size_t)j2 * N1 + k1) * 10));
    frn_tile_stages(frn_lds, p.t2, tab + (size_t)frn_off_loc2(p) * 10);
    for (u32 k2 = threadIdx.x; k2 < N2; k2 += FRN_TILE_THREADS)
        frn_pass_b_out(data, frn_tile_load(frn_lds, N2, k2), tab, p, k1, k2, coset_out);
}

// the resident table block of (log_n, direction), built on first use on `st`
static int32_t frn_tables(zklc_ctx *ctx, hipStream_t st, const frn_plan &p, bool inverse, const i32 **out) {
    static std::mutex tab_mutex;                // the build is serialised and complete before the pointer is published
    std::lock_guard<std::mutex> lk(tab_mutex);
    void *&slot = ctx->fr_ntt_tab[inverse ? 1 : 0][p.log_n];
    if (!slot) {
        void *fresh = nullptr;
        ZKLC_HIP(ctx, hipMalloc(&fresh, (size_t)frn_table_elems(p) * 40));
        hipLaunchKernelGGL(frn_table_consts_kernel, dim3(1), dim3(64), 0, st, (i32 *)fresh, p.log_n, (u32)inverse);
        u32 ne = frn_table_elems(p) - 4;
        hipLaunchKernelGGL(frn_table_entries_kernel, dim3((ne + 255) / 256), dim3(256), 0, st, (i32 *)fresh, p);
        ZKLC_HIP(ctx, hipGetLastError());
        // other streams of this context may use the block right after this call returns
        ZKLC_HIP(ctx, zklc_stream_wait(st));
        slot = fresh;
    }
    *out = (const i32 *)slot;
    return ZKLC_OK;
}
void zklc_bn254_fr_ntt_fini(zklc_ctx *ctx) {
    for (auto &dir : ctx->fr_ntt_tab)
        for (auto &t : dir)
            if (t) {
                (void)hipFree(t);
                t = nullptr;
            }
}

static int32_t frn_two_pass(zklc_ctx *ctx, hipStream_t st, uint64_t *d_data, uint32_t log_n, bool inverse, uint32_t coset, i32 *mid) {
    ZKLC_HIP(ctx, zklc_once_per_device([] {
        hipError_t e = hipFuncSetAttribute((const void *)frn_pass_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute((const void *)frn_pass_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    }));
    frn_plan p = frn_make_plan(log_n);
    const i32 *tab;
    int32_t rc = frn_tables(ctx, st, p, inverse, &tab);
    if (rc) return rc;
    const u32 N1 = 1u << p.t1, N2 = 1u << p.t2;
    hipLaunchKernelGGL(frn_pass_a_kernel, dim3(N2), dim3(FRN_TILE_THREADS), (size_t)N1 * 40, st, (const u64 *)d_data, mid, tab, p,
                       (u32)(coset && !inverse));
    hipLaunchKernelGGL(frn_pass_b_kernel, dim3(N1), dim3(FRN_TILE_THREADS), (size_t)N2 * 40, st, (const i32 *)mid, d_data, tab, p,
                       (u32)(coset && inverse));
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" int32_t zklc_bn254_fr_mul_sub_scale_dev(zklc_ctx *ctx, void *stream, uint64_t *d_a, const uint64_t *d_b, const uint64_t *d_c,
                                                   const uint64_t *scale, uint64_t n) {
    if (!ctx || !d_a || !d_b || !d_c || !scale) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = zklc_pick_stream(ctx, stream);
    if (n)
        hipLaunchKernelGGL(frn_mul_sub_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_a, d_b, d_c, scale[0], scale[1], scale[2],
                           scale[3], n);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" uint64_t zklc_bn254_fr_ntt_workspace_bytes(uint32_t log_n) { return ((uint64_t)40 << log_n) + ((uint64_t)20 << log_n) + 512; }

extern "C" int32_t zklc_bn254_fr_ntt_dev(zklc_ctx *ctx, void *stream, uint64_t *d_data, uint32_t log_n, uint32_t flags, uint32_t coset,
                                         void *d_workspace, uint64_t workspace_bytes) {
    if (!ctx || !d_data || log_n > 28 || !d_workspace || workspace_bytes < zklc_bn254_fr_ntt_workspace_bytes(log_n))
        return ZKLC_ERR_INVALID_ARG;
    bool inverse = flags & ZKLC_NTT_INVERSE, in_br = flags & ZKLC_NTT_IN_BITREV, out_br = flags & ZKLC_NTT_OUT_BITREV;
    if (in_br && out_br) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = zklc_pick_stream(ctx, stream);
    u32 n = 1u << log_n, half = n >> 1;
    i32 *work = (i32 *)d_workspace;
    // A/B switch ZKLC_FR_NTT=stages keeps the one-launch-per-stage path at every size
    static const bool two_pass = [] { const char *v = getenv("ZKLC_FR_NTT"); return !(v && v[0] == 's'); }();
    if (two_pass && !in_br && !out_br && log_n >= FRN_FAST_MIN_LOG && log_n <= FRN_FAST_MAX_LOG)
        return frn_two_pass(ctx, st, d_data, log_n, inverse, coset, work);
    i32 *tw = work + (size_t)n * 10;
    i32 *consts = tw + (size_t)(half ? half : 1) * 10;
    hipLaunchKernelGGL(frn_consts_kernel, dim3(1), dim3(64), 0, st, consts, log_n);
    if (half) hipLaunchKernelGGL(frn_twiddle_kernel, dim3((half + 255) / 256), dim3(256), 0, st, tw, log_n, (u32)inverse, half);
    // DIT needs bit-reversed input: gather through the bit reversal unless the caller already supplies that order.
    // A bit-reversed OUTPUT is produced by running the natural-order transform and scattering through the reversal.
    hipLaunchKernelGGL(frn_load_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const u64 *)d_data, work, (const i32 *)consts, log_n,
                       (u32)!in_br, (u32)(coset && !inverse));
    for (u32 s = 0; s < log_n; s++)
        hipLaunchKernelGGL(frn_stage_kernel, dim3((half + 255) / 256), dim3(256), 0, st, work, (const i32 *)tw, log_n, s);
    hipLaunchKernelGGL(frn_store_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const i32 *)work, d_data, (const i32 *)consts, log_n,
                       (u32)out_br, (u32)inverse, coset);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" int32_t zklc_bn254_fr_ntt(zklc_ctx *ctx, uint64_t *data, uint32_t log_n, uint32_t flags, uint32_t coset) {
    if (!ctx || !data || log_n > 28) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t bytes = (size_t)32 << log_n;
    uint64_t wb = zklc_bn254_fr_ntt_workspace_bytes(log_n);
    void *d, *w;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, bytes, &d))) return rc;
    if ((rc = zklc_stage(ctx, 1, wb, &w))) return rc;
    ZKLC_HIP(ctx, hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = zklc_bn254_fr_ntt_dev(ctx, ctx->stream, (uint64_t *)d, log_n, flags, coset, w, wb))) return rc;
    ZKLC_HIP(ctx, zklc_readback_async(data, d, bytes, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}
