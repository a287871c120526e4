// BN254 pairing-product check on gfx950 + C ABI (bn254_pairing.cuh).  Throughput form: one lane per check (k Miller loops folded
// into one Fp12 accumulator, then one final exponentiation) -- SURVEY 8d: "batch for throughput".  Latency form (batch <= 2048): a
// workgroup per check, one lane per pair.
#include "bn254_pairing.cuh"
#include "zklc_internal.h"

__global__ void __launch_bounds__(64)
bn254_pairing_check_kernel(const u64 *__restrict__ g1, const u64 *__restrict__ g2, u32 k, u32 batch, u32 *__restrict__ is_one,
                           u64 *__restrict__ gt_out) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    fp12 f = f12_one();
    for (u32 i = 0; i < k; i++) {
        const u32 *p = reinterpret_cast<const u32 *>(g1 + ((size_t)b * k + i) * 8);
        const u32 *q = reinterpret_cast<const u32 *>(g2 + ((size_t)b * k + i) * 16);
        u32 w1[16], w2[32], z1 = 0, z2 = 0;
        for (int j = 0; j < 16; j++) z1 |= (w1[j] = p[j]);
        for (int j = 0; j < 32; j++) z2 |= (w2[j] = q[j]);
        if (!z1 || !z2) continue;  // e(O, Q) = e(P, O) = 1
        fp xp = fp_reduce(fp_from_gnark(w1)), yp = fp_reduce(fp_from_gnark(w1 + 8));
        fp2 xq = fp2_reduce(fp2_from_gnark(w2)), yq = fp2_reduce(fp2_from_gnark(w2 + 16));
        bn_miller_loop(f, xp, yp, xq, yq);
    }
    f = bn_final_exponentiation(f);
    is_one[b] = f12_is_one(f);
    if (gt_out) {
        u32 w[96];
        f12_to_gnark(w, f);
        for (int j = 0; j < 48; j++) gt_out[(size_t)b * 48 + j] = (u64)w[2 * j] | ((u64)w[2 * j + 1] << 32);
    }
}

// Latency form for small batches (one Groth16 verification = ONE check of k = 4 pairings): a 64-lane workgroup per check, lane j
// runs the Miller loop of pair j (the k loops are independent: they were run one after the other by one lane), the k values are
// multiplied through LDS and lane 0 does the final exponentiation.  Throughput form above for large batches.
#define BN_SPLIT_MAX_BATCH 2048u
ZKLC_D void f12_to_lds(i32 *dst, const fp12 &a) {
    const fp2 *x[6] = {&a.c0.b0, &a.c0.b1, &a.c0.b2, &a.c1.b0, &a.c1.b1, &a.c1.b2};
    for (int i = 0; i < 6; i++)
        for (int k = 0; k < 10; k++) {
            dst[20 * i + k] = x[i]->c0.v[k];
            dst[20 * i + 10 + k] = x[i]->c1.v[k];
        }
}
ZKLC_D fp12 f12_from_lds(const i32 *src) {
    fp12 a;
    fp2 *x[6] = {&a.c0.b0, &a.c0.b1, &a.c0.b2, &a.c1.b0, &a.c1.b1, &a.c1.b2};
    for (int i = 0; i < 6; i++)
        for (int k = 0; k < 10; k++) {
            x[i]->c0.v[k] = src[20 * i + k];
            x[i]->c1.v[k] = src[20 * i + 10 + k];
        }
    return a;
}
__global__ void __launch_bounds__(64)
bn254_pairing_check_split_kernel(const u64 *__restrict__ g1, const u64 *__restrict__ g2, u32 k, u32 *__restrict__ is_one,
                                 u64 *__restrict__ gt_out) {
    __shared__ i32 lds[64 * 120];
    const u32 b = blockIdx.x, lane = threadIdx.x;
    fp12 f = f12_one();
    for (u32 i = lane; i < k; i += 64) {                       // k <= 64 in practice: one pair per lane
        const u32 *p = reinterpret_cast<const u32 *>(g1 + ((size_t)b * k + i) * 8);
        const u32 *q = reinterpret_cast<const u32 *>(g2 + ((size_t)b * k + i) * 16);
        u32 w1[16], w2[32], z1 = 0, z2 = 0;
        for (int j = 0; j < 16; j++) z1 |= (w1[j] = p[j]);
        for (int j = 0; j < 32; j++) z2 |= (w2[j] = q[j]);
        if (!z1 || !z2) continue;  // e(O, Q) = e(P, O) = 1
        fp xp = fp_reduce(fp_from_gnark(w1)), yp = fp_reduce(fp_from_gnark(w1 + 8));
        fp2 xq = fp2_reduce(fp2_from_gnark(w2)), yq = fp2_reduce(fp2_from_gnark(w2 + 16));
        bn_miller_loop(f, xp, yp, xq, yq);
    }
    u32 width = 1;
    while (width < k && width < 64) width <<= 1;
    for (u32 stride = width >> 1; stride >= 1; stride >>= 1) {
        if (lane >= stride && lane < 2 * stride) f12_to_lds(lds + (lane - stride) * 120, f);
        __syncthreads();
        if (lane < stride) f = f12_mul(f, f12_from_lds(lds + lane * 120));
        __syncthreads();
    }
    if (lane != 0) return;
    f = bn_final_exponentiation(f);
    is_one[b] = f12_is_one(f);
    if (gt_out) {
        u32 w[96];
        f12_to_gnark(w, f);
        for (int j = 0; j < 48; j++) gt_out[(size_t)b * 48 + j] = (u64)w[2 * j] | ((u64)w[2 * j + 1] << 32);
    }
}

extern "C" int32_t zklc_bn254_pairing_check_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_g1, const uint64_t *d_g2, uint32_t k,
                                                uint32_t batch, uint32_t *d_is_one, uint64_t *d_gt_out) {
    if (!ctx || !d_is_one || (batch && k && (!d_g1 || !d_g2))) return ZKLC_ERR_INVALID_ARG;
    if (batch == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    if (batch <= BN_SPLIT_MAX_BATCH && k > 1)
        hipLaunchKernelGGL(bn254_pairing_check_split_kernel, dim3(batch), dim3(64), 0, zklc_pick_stream(ctx, stream), d_g1, d_g2, k,
                           d_is_one, d_gt_out);
    else
        hipLaunchKernelGGL(bn254_pairing_check_kernel, dim3((batch + 63) / 64), dim3(64), 0, zklc_pick_stream(ctx, stream), d_g1, d_g2, k,
                           batch, d_is_one, d_gt_out);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" int32_t zklc_bn254_pairing_check(zklc_ctx *ctx, const uint64_t *g1, const uint64_t *g2, uint32_t k, uint32_t batch,
                                            uint32_t *is_one, uint64_t *gt_out) {
    if (!ctx || !is_one || (batch && k && (!g1 || !g2))) return ZKLC_ERR_INVALID_ARG;
    if (batch == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t n = (size_t)batch * k;
    void *d1, *d2, *dr, *dg;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, n * 64 + 64, &d1))) return rc;
    if ((rc = zklc_stage(ctx, 1, n * 128 + 128, &d2))) return rc;
    if ((rc = zklc_stage(ctx, 2, (size_t)batch * 4, &dr))) return rc;
    if ((rc = zklc_stage(ctx, 3, (size_t)batch * 384, &dg))) return rc;
    if (n) {
        ZKLC_HIP(ctx, hipMemcpyAsync(d1, g1, n * 64, hipMemcpyHostToDevice, ctx->stream));
        ZKLC_HIP(ctx, hipMemcpyAsync(d2, g2, n * 128, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = zklc_bn254_pairing_check_dev(ctx, ctx->stream, (const uint64_t *)d1, (const uint64_t *)d2, k, batch, (uint32_t *)dr,
                                           gt_out ? (uint64_t *)dg : nullptr)))
        return rc;
    ZKLC_HIP(ctx, zklc_readback_async(is_one, dr, (size_t)batch * 4, ctx->stream));
    if (gt_out) ZKLC_HIP(ctx, zklc_readback_async(gt_out, dg, (size_t)batch * 384, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}
