// Batched Ed25519 verification + SHA-512 kernels (gfx950) and their C ABI.
// The verify kernel itself lives in ed25519_kernels.inc (instantiated by
// ed25519_v1.hip: 8-entry per-lane table in LDS, field operations inlined -- the
// fastest of the four variants measured in round 1, profiles/r01_ed25519_variants_v2.txt);
// this TU holds the base-table and SHA-512 kernels and the C ABI entry points.
#include "ed25519_verify.cuh"
#include "zklc_internal.h"
#include <stdlib.h>

void zklc_ed_launch_v1(hipStream_t, const uint8_t *, const uint8_t *, const uint8_t *, u32, u32, u32, const void *, uint8_t *);

__global__ void __launch_bounds__(128) ed25519_base_table_kernel(ge_niels *tab) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < ZKLC_ED_BTABLE) tab[j] = ed25519_base_table_entry(j + 1);
}

__global__ void __launch_bounds__(256)
sha512_batch_kernel(const uint8_t *__restrict__ in, u32 stride, u32 len, u32 n, uint8_t *__restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 h[8];
    sha512_hash_msg(in + (size_t)i * stride, len, h);
    // big-endian digest bytes; two 16-byte stores per 32 bytes
    uint4 *o = reinterpret_cast<uint4 *>(out + (size_t)i * 64);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        u64 x = __builtin_bswap64(h[2 * q]), y = __builtin_bswap64(h[2 * q + 1]);
        uint4 v;
        v.x = (u32)x; v.y = (u32)(x >> 32); v.z = (u32)y; v.w = (u32)(y >> 32);
        o[q] = v;
    }
}

int32_t zklc_ed25519_init(zklc_ctx *ctx) {
    ctx->ed_variant = 1;
    ZKLC_HIP(ctx, hipMalloc(&ctx->ed_btab, sizeof(ge_niels) * ZKLC_ED_BTABLE));
    hipLaunchKernelGGL(ed25519_base_table_kernel, dim3(1), dim3(128), 0, ctx->stream, (ge_niels *)ctx->ed_btab);
    ZKLC_HIP(ctx, hipGetLastError());
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}

void zklc_ed25519_fini(zklc_ctx *ctx) {
    if (ctx->ed_btab) (void)hipFree(ctx->ed_btab);
    ctx->ed_btab = nullptr;
}

extern "C" int32_t zklc_ed25519_verify_batch_dev(zklc_ctx *ctx, void *stream, const uint8_t *d_pks, const uint8_t *d_sigs,
                                                 const uint8_t *d_msgs, uint32_t msg_len, uint32_t msg_stride, uint32_t n,
                                                 uint8_t *d_ok) {
    if (!ctx) return ZKLC_ERR_INVALID_ARG;
    if (n == 0) return ZKLC_OK;
    if (!d_pks || !d_sigs || !d_ok || (msg_len && !d_msgs)) return ZKLC_ERR_INVALID_ARG;
    if (msg_stride != 0 && msg_stride < msg_len) return ZKLC_ERR_INVALID_ARG;
    if (((uintptr_t)d_pks | (uintptr_t)d_sigs) & 15) return ZKLC_ERR_INVALID_ARG;  // 16-byte vector loads
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    zklc_ed_launch_v1(zklc_pick_stream(ctx, stream), d_pks, d_sigs, d_msgs, msg_len, msg_stride, n, ctx->ed_btab, d_ok);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" int32_t zklc_ed25519_verify_batch(zklc_ctx *ctx, const uint8_t *pks, const uint8_t *sigs, const uint8_t *msgs,
                                             uint32_t msg_len, uint32_t msg_stride, uint32_t n, uint8_t *ok) {
    if (!ctx) return ZKLC_ERR_INVALID_ARG;
    if (n == 0) return ZKLC_OK;
    if (!pks || !sigs || !ok || (msg_len && !msgs)) return ZKLC_ERR_INVALID_ARG;
    if (msg_stride != 0 && msg_stride < msg_len) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t msg_bytes = msg_stride ? (size_t)msg_stride * n : msg_len;
    void *dpk, *dsg, *dmsg, *dok;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, (size_t)n * 32, &dpk))) return rc;
    if ((rc = zklc_stage(ctx, 1, (size_t)n * 64, &dsg))) return rc;
    if ((rc = zklc_stage(ctx, 2, msg_bytes ? msg_bytes : 1, &dmsg))) return rc;
    if ((rc = zklc_stage(ctx, 3, n, &dok))) return rc;
    ZKLC_HIP(ctx, hipMemcpyAsync(dpk, pks, (size_t)n * 32, hipMemcpyHostToDevice, ctx->stream));
    ZKLC_HIP(ctx, hipMemcpyAsync(dsg, sigs, (size_t)n * 64, hipMemcpyHostToDevice, ctx->stream));
    if (msg_bytes) ZKLC_HIP(ctx, hipMemcpyAsync(dmsg, msgs, msg_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = zklc_ed25519_verify_batch_dev(ctx, ctx->stream, (const uint8_t *)dpk, (const uint8_t *)dsg, (const uint8_t *)dmsg,
                                       msg_len, msg_stride, n, (uint8_t *)dok);
    if (rc) return rc;
    ZKLC_HIP(ctx, zklc_readback_async(ok, dok, n, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}

extern "C" int32_t zklc_sha512_batch_dev(zklc_ctx *ctx, void *stream, const uint8_t *d_in, uint32_t stride, uint32_t len,
                                         uint32_t n, uint8_t *d_out) {
    if (!ctx) return ZKLC_ERR_INVALID_ARG;
    if (n == 0) return ZKLC_OK;
    if (!d_out || (len && !d_in) || stride < len) return ZKLC_ERR_INVALID_ARG;
    if ((uintptr_t)d_out & 15) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(sha512_batch_kernel, dim3((n + 255) / 256), dim3(256), 0, zklc_pick_stream(ctx, stream), d_in,
                       stride, len, n, d_out);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" int32_t zklc_sha512_batch(zklc_ctx *ctx, const uint8_t *in, uint32_t stride, uint32_t len, uint32_t n,
                                     uint8_t *out) {
    if (!ctx) return ZKLC_ERR_INVALID_ARG;
    if (n == 0) return ZKLC_OK;
    if (!out || (len && !in) || stride < len) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t in_bytes = (size_t)stride * (n - 1) + len;
    void *din, *dout;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, in_bytes ? in_bytes : 1, &din))) return rc;
    if ((rc = zklc_stage(ctx, 1, (size_t)n * 64, &dout))) return rc;
    if (in_bytes) ZKLC_HIP(ctx, hipMemcpyAsync(din, in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = zklc_sha512_batch_dev(ctx, ctx->stream, (const uint8_t *)din, stride, len, n, (uint8_t *)dout);
    if (rc) return rc;
    ZKLC_HIP(ctx, zklc_readback_async(out, dout, (size_t)n * 64, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}
