// BN254 pairing-product check on gfx950 + C ABI (bn254_pairing.cuh): one lane per check (k Miller loops sharing the
// squarings are folded into one Fp12 accumulator, then one final exponentiation).  A single check is latency-bound
// (~2e7 dependent instructions); throughput comes from batching checks across lanes ("batch for throughput").
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

extern "C" int32_t zklc_bn254_pairing_check_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_g1, const uint64_t *d_g2, uint32_t k,
                                                uint32_t batch, uint32_t *d_is_one, uint64_t *d_gt_out) {
    if (!ctx || !d_is_one || (batch && k && (!d_g1 || !d_g2))) return ZKLC_ERR_INVALID_ARG;
    if (batch == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
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
    ZKLC_HIP(ctx, hipMemcpyAsync(is_one, dr, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (gt_out) ZKLC_HIP(ctx, hipMemcpyAsync(gt_out, dg, (size_t)batch * 384, hipMemcpyDeviceToHost, ctx->stream));
    ZKLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKLC_OK;
}
