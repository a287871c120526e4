// Context management for libzklc_mi355.so (C ABI declared in include/zklc.h).
#include "zklc_internal.h"
#include <cstdlib>
#include <new>

int32_t zklc_stage(zklc_ctx *ctx, int slot, size_t bytes, void **out) {
    zklc_devbuf &b = ctx->stage[slot];
    if (b.cap < bytes) {
        if (b.p) {
            ZKLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    {
        // Round 5: a proving thread that waits for its stream (~10 waits per proof) was measured at 1.00 host cores busy -- the
        // runtime's default wait spins (profiles/r05d_host_cpu_probe.txt): six cores per rank with the pipeline's six threads.
        // ZKLC_BLOCKING_WAIT=1 asks the device for blocking waits (the thread sleeps on the completion interrupt: 0.14 cores busy at
        // +1 % wall time, profiles/r05e_host_cpu_probe_blocking.txt).  OPT-IN: the flag is device-wide -- it also changes how the
        // caller's own runtime calls wait (torch's allocator, hipFree) -- and the one full bench run made with it as the default did
        // not get past torch.cuda.empty_cache() (profiles/r05h_*); the GPU suite, smoke() and whole block proofs passed with it.
        // What separates the two: bench.py had run torch kernels on the device BEFORE its first zklc_init flipped the wait mode,
        // the passing programs created their context first.  Like cudaSetDeviceFlags, the mode belongs at process start: use
        // ZKLC_BLOCKING_WAIT=1 only in a process whose first GPU call is zklc_init (a Rust / Go host; the pipeline's own ranks).
        static const bool blocking = getenv("ZKLC_BLOCKING_WAIT") && getenv("ZKLC_BLOCKING_WAIT")[0] == '1';
        if (blocking) {
            (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
            (void)hipGetLastError();
        }
    }
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

extern "C" int32_t zklc_synchronize(zklc_ctx *ctx) {
    if (!ctx) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    ZKLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKLC_OK;
}
