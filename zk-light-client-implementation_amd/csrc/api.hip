// Context management for libzklc_mi355.so (C ABI declared in include/zklc.h).
#include "zklc_internal.h"
#include <cstdlib>
#include <new>

int32_t zklc_stage(zklc_ctx *ctx, int slot, size_t bytes, void **out) {
    zklc_devbuf &b = ctx->stage[slot];
    if (b.cap < bytes) {
        if (b.p) {
            ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
            ZKLC_HIP(ctx, hipFree(b.p));
            b.p = nullptr;
            b.cap = 0;
        }
        size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes;
        ZKLC_HIP(ctx, hipMalloc(&b.p, cap));
        b.cap = cap;
    }
    *out = b.p;
    return ZKLC_OK;
}

extern "C" uint32_t zklc_abi_version(void) { return 2; }   // 2: circuit container (round 6)

extern "C" const char *zklc_strerror(int32_t code) {
    switch (code) {
        case ZKLC_OK: return "ok";
        case ZKLC_ERR_INVALID_ARG: return "invalid argument";
        case ZKLC_ERR_OOM: return "out of device memory";
        case ZKLC_ERR_HIP: return "HIP runtime error (see zklc_last_hip_error)";
        case ZKLC_ERR_NO_DEVICE: return "no usable gfx950 device";
        case ZKLC_ERR_IO: return "container file could not be opened or written";
        case ZKLC_ERR_FORMAT: return "not a valid circuit container (magic, version, size, checksum or inconsistent sections)";
        case ZKLC_ERR_NOT_FOUND: return "no such section in the container";
        default: return "unknown zklc status";
    }
}

extern "C" const char *zklc_last_hip_error(zklc_ctx *ctx) { return ctx ? ctx->last_err.c_str() : ""; }

extern "C" int32_t zklc_init(zklc_ctx **out, int32_t device_id) { return zklc_init_priority(out, device_id, 0); }

extern "C" int32_t zklc_init_priority(zklc_ctx **out, int32_t device_id, int32_t high_priority) {
    if (!out || device_id < 0) return ZKLC_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZKLC_ERR_NO_DEVICE;
    if (device_id >= count) return ZKLC_ERR_INVALID_ARG;
    zklc_ctx *ctx = new (std::nothrow) zklc_ctx();
    if (!ctx) return ZKLC_ERR_OOM;
    ctx->device = device_id;
    int32_t rc = ZKLC_OK;
    auto fail = [&](int32_t code) {
        zklc_destroy(ctx);
        return code;
    };
    if (hipSetDevice(device_id) != hipSuccess) return fail(ZKLC_ERR_NO_DEVICE);
    // Waiting.  Rounds 1-5 left a proving thread at 1.00 host cores busy; round 5 offered hipDeviceScheduleBlockingSync behind
    // ZKLC_BLOCKING_WAIT=1, a DEVICE-WIDE mode that also changed how the caller's own runtime calls wait (and the one bench run with it
    // as the default hung in torch's allocator).  Round 6 found the spinning itself: hipMemcpyAsync to / from pageable host memory
    // waits for the stream inside the call.  Every transfer of the per-proof paths now goes through page-locked staging
    // (plonky2_prover.hip `p2_pin`, plonky2_witness_dev.hip `h_pin`) and every wait through zklc_stream_wait (an event created with
    // hipEventBlockingSync): the library never touches the device's scheduling flags.
    int prio_low = 0, prio_high = 0;   // numerically lower = higher priority
    if (high_priority && hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess) prio_high = 0;
    if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, high_priority ? prio_high : 0) != hipSuccess)
        return fail(ZKLC_ERR_HIP);
    if ((rc = zklc_ed25519_init(ctx))) return fail(rc);
    *out = ctx;
    return ZKLC_OK;
}

extern "C" void zklc_destroy(zklc_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    zklc_ed25519_fini(ctx);
    zklc_gl_fini(ctx);
    zklc_bn254_fr_ntt_fini(ctx);
    for (auto &b : ctx->stage)
        if (b.p) (void)hipFree(b.p);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" void *zklc_stream(zklc_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// Device memory for hosts that bind nothing but this ABI (tests/c_abi/prove_from_file.c; a Rust / Go shim need not link HIP)
extern "C" int32_t zklc_device_alloc(zklc_ctx *ctx, uint64_t bytes, void **d_out) {
    if (!ctx || !d_out) return ZKLC_ERR_INVALID_ARG;
    *d_out = nullptr;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    void *p = nullptr;
    ZKLC_HIP(ctx, hipMalloc(&p, bytes ? bytes : 8));
    hipError_t e = hipMemsetAsync(p, 0, bytes ? bytes : 8, ctx->stream);
    if (e == hipSuccess) e = zklc_stream_wait(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(p);
        ctx->last_err = std::string("zklc_device_alloc: ") + hipGetErrorString(e);
        return ZKLC_ERR_HIP;
    }
    *d_out = p;
    return ZKLC_OK;
}

extern "C" int32_t zklc_device_free(zklc_ctx *ctx, void *d_ptr) {
    if (!ctx) return ZKLC_ERR_INVALID_ARG;
    if (!d_ptr) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    ZKLC_HIP(ctx, hipFree(d_ptr));
    return ZKLC_OK;
}

extern "C" int32_t zklc_device_copy(zklc_ctx *ctx, void *dst, const void *src, uint64_t bytes, int32_t to_host) {
    if (!ctx || (bytes && (!dst || !src))) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    ZKLC_HIP(ctx, hipMemcpyAsync(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}

extern "C" int32_t zklc_synchronize(zklc_ctx *ctx) {
    if (!ctx) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}
