// Internal context shared by the translation units of libzklc_mi355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <array>
#include <map>
#include <mutex>
#include <string>
#include "../../include/zklc.h"

#define ZKLC_GL_MAX_LOG 24

struct zklc_devbuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct zklc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_err;
    // Ed25519: 128 affine-niels multiples of the base point (12 KiB)
    void *ed_btab = nullptr;
    int ed_variant = 0;  // index into the compiled verify-kernel variants
    // Goldilocks: twiddle tables w^i (i < 2^(logn-1)) per transform size, forward and inverse
    void *gl_tw_fwd[ZKLC_GL_MAX_LOG + 1] = {};
    void *gl_tw_inv[ZKLC_GL_MAX_LOG + 1] = {};
    // staged tables (per stage s: w^(j << s) contiguous), used by the radix-8 passes
    void *gl_tws_fwd[ZKLC_GL_MAX_LOG + 1] = {};
    void *gl_tws_inv[ZKLC_GL_MAX_LOG + 1] = {};
    // per-element group twiddles of the shift-twiddle passes (goldilocks_ntt_group.cuh): blocks w^((bitrev_g(m) J) << s') per group
    void *gl_twg_fwd[ZKLC_GL_MAX_LOG + 1] = {};
    void *gl_twg_inv[ZKLC_GL_MAX_LOG + 1] = {};
    // two-level table of coset-shift powers (shift^j = hi[j >> 10] * lo[j & 1023])
    void *gl_scale_hi = nullptr, *gl_scale_lo = nullptr;
    uint64_t gl_scale_shift = 0;
    int gl_scale_hi_n = 0;
    // BN254 Fr NTT: resident table block per (direction, log n) of the two-pass transform (bn254_fr_ntt_tile.cuh)
    void *fr_ntt_tab[2][29] = {};
    // grow-only staging buffers for the host-pointer entry points (slot 7 = kernel scratch)
    zklc_devbuf stage[8];
    // fixed-base MSM tables this context has built or validated: device address -> header words (bn254_msm.hip), so that the
    // *_msm_fixed_dev entry points stay enqueue-only (no device read, no null-stream copy) after a table's first use
    std::map<const void *, std::array<uint32_t, 5>> msm_tables;
    std::mutex msm_tables_mu;
};

#define ZKLC_HIP(ctx, call)                                           \
    do {                                                              \
        hipError_t e__ = (call);                                      \
        if (e__ != hipSuccess) {                                      \
            (ctx)->last_err = std::string(#call) + ": " + hipGetErrorString(e__); \
            return e__ == hipErrorOutOfMemory ? ZKLC_ERR_OOM : ZKLC_ERR_HIP;      \
        }                                                             \
    } while (0)

// Waiting for a stream (every host wait of the library goes through here).
// Measured on this runtime (profiles/r06b_host_cpu_probe.txt, r05d): hipStreamSynchronize spins, and hipEventSynchronize on an event
// created with hipEventBlockingSync spins as well -- a proving thread sat at 1.00 host cores whatever the event's flags, also after
// every transfer had been moved to page-locked staging; only the DEVICE-wide hipDeviceScheduleBlockingSync made the runtime sleep
// (round 5's opt-in, removed: a library must not flip a device-wide mode under its caller).  Round 6 therefore does the waiting
// itself: record an event, query it for ~20 us (the waits inside a small proof end within that), then sleep between queries --
// 10 us growing to 60 us, the thread's timer slack set to 1 us so that the sleeps are what they say.  The thread is asleep for all
// but a few microseconds per query; the price is at most one sleep interval of latency per wait.
//   ZKLC_WAIT=poll (default) | event (hipEventSynchronize on a blocking-sync event) | spin (hipStreamSynchronize)      [A/B]
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <time.h>
inline int zklc_wait_mode() {
    static const int mode = [] {
        const char *e = getenv("ZKLC_WAIT");
        if (e && !strcmp(e, "spin")) return 0;
        if (e && !strcmp(e, "event")) return 1;
        return 2;
    }();
    return mode;
}
inline hipError_t zklc_stream_wait(hipStream_t st) {
    const int mode = zklc_wait_mode();
    if (mode == 0) return hipStreamSynchronize(st);
    static thread_local hipEvent_t ev = nullptr;
    static thread_local int ev_dev = -1;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!ev || ev_dev != dev) {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming);
        if (e != hipSuccess) return e;
        ev_dev = dev;
        (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);      // this thread's nanosleep wakes within ~1 us of its deadline
    }
    e = hipEventRecord(ev, st);
    if (e != hipSuccess) return e;
    if (mode == 1) return hipEventSynchronize(ev);
    auto now_ns = [] {
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return (long long)t.tv_sec * 1000000000LL + t.tv_nsec;
    };
    const long long t0 = now_ns();
    long sleep_ns = 10000;
    for (;;) {
        e = hipEventQuery(ev);
        if (e != hipErrorNotReady) break;
        if (now_ns() - t0 < 20000) continue;                    // short waits: stay on the core
        timespec ts = {0, sleep_ns};
        nanosleep(&ts, nullptr);
        if (sleep_ns < 60000) sleep_ns += sleep_ns / 2;          // 10 -> 60 us: r06c measured +185 us per wait with a 200 us cap
    }
    (void)hipGetLastError();                                    // hipErrorNotReady of the queries is not an error of the caller's
    return e;
}

// Device -> host read-back of a result the host is about to wait for.  Enqueued right behind the kernels that produce it, the copy
// command sits in the runtime's DMA queue until those kernels have ended -- and copies of OTHER streams that the runtime put into the
// same queue wait behind it (round 6, kernel trace cut by hardware queue: two of the three Ed25519 prover streams stood still for
// 0.6-1.0 s, between their proof-of-work kernel and the gather of the query openings, exactly as long as the witness producer's
// batch -- 2 300 small kernels with the read-back of its public inputs enqueued behind them -- was running; the third prover
// stream, on another queue, went on).  So: wait for the stream FIRST, then enqueue the copy (it runs at once), then the caller waits
// for it as before.  ZKLC_SETTLED_COPIES=0 restores the enqueue-behind-the-kernels form (A/B).
inline bool zklc_settled_copies() {
    static const bool on = !(getenv("ZKLC_SETTLED_COPIES") && getenv("ZKLC_SETTLED_COPIES")[0] == '0');
    return on;
}
inline hipError_t zklc_readback_async(void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (zklc_settled_copies()) {
        hipError_t e = zklc_stream_wait(st);
        if (e != hipSuccess) return e;
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
}

// returns a device buffer of at least `bytes` in slot `slot`
int32_t zklc_stage(zklc_ctx *ctx, int slot, size_t bytes, void **out);
// *_dev entry points launch on exactly the hipStream_t they are given (NULL = the legacy default stream)
inline hipStream_t zklc_pick_stream(zklc_ctx *, void *s) { return (hipStream_t)s; }


// hipFuncSetAttribute (e.g. more than 64 KiB of dynamic LDS) applies to the CURRENT device only: run `set` once per device of this
// process, not once per process (a process that drives two GPUs would otherwise launch on the second one without the attribute)
#include <mutex>
template <class Fn>
inline hipError_t zklc_once_per_device(Fn set) {
    static std::mutex m;
    static bool done[64];
    static hipError_t res[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return set();
    std::lock_guard<std::mutex> lk(m);
    if (!done[dev]) {
        res[dev] = set();
        done[dev] = true;
    }
    return res[dev];
}

// subsystem initialisers (called by zklc_init)
int32_t zklc_ed25519_init(zklc_ctx *ctx);
void zklc_ed25519_fini(zklc_ctx *ctx);
void zklc_gl_fini(zklc_ctx *ctx);
void zklc_bn254_fr_ntt_fini(zklc_ctx *ctx);

// Merkle commit with an explicit leaf layout: element q of leaf i at d_mat[q * stride + i * leaf_stride]
// (poly-major LDE matrices: leaf_stride 1; row-major FRI leaves: stride 1, leaf_stride = width)
// values -> coefficients (natural order) from `d_src` into `d_dst` (d_src is not modified; the first pass reads it, the rest runs in
// place on d_dst): the prover keeps the wire values and needs the coefficients -- no device-to-device copy in front of an in-place
// transform
int32_t zklc_gl_intt_copy_dev(zklc_ctx *ctx, hipStream_t st, const uint64_t *d_src, uint64_t *d_dst, uint32_t log_n, uint32_t batch);
int32_t zklc_gl_merkle_commit_strided(zklc_ctx *ctx, hipStream_t st, const uint64_t *d_mat, uint64_t stride, uint64_t leaf_stride,
                                      uint32_t log_leaves, uint32_t width, uint32_t cap_height, uint64_t *d_tree);
int32_t zklc_bn254_merkle_commit_strided(zklc_ctx *ctx, hipStream_t st, const uint64_t *d_mat, uint64_t stride, uint64_t leaf_stride,
                                         uint32_t log_leaves, uint32_t width, uint32_t cap_height, uint64_t *d_tree);
