// gfx950 micro-benchmark of the BN254 field / curve code AS COMPILED (csrc/mont26_impl.inc, bn254_ec.cuh), without memory traffic:
// what fraction of the integer issue ceiling (tools/ubench/valu_ubench) does a chain of fp_mul / ec_add_affine reach at 1, 2, 4
// waves per SIMD?  Separates "the bucket kernel waits for its gathers" from "the arithmetic itself does not issue every slot".
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zk-light-client-implementation_amd/csrc tools/ubench/fp_ubench.hip -o tools/ubench/fp_ubench
#include "bn254_msm_lane.cuh"
#include <stdio.h>
#include <vector>

#define FP_MUL_INSTR 294.0      // VALU instructions of one compiled fp_mul (hipcc -S, round 3)

template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_fpmul(i32 *io, int n) {
    fp a, b;
    for (int i = 0; i < 10; i++) {
        a.v[i] = io[threadIdx.x * 20 + i];
        b.v[i] = io[threadIdx.x * 20 + 10 + i];
    }
#pragma unroll 1
    for (int it = 0; it < n; it++) a = fp_mul(a, b);
    if (a.v[0] == 0x7fffffff) io[0] = a.v[1];
}
template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_fpmul2(i32 *io, int n) {       // two independent chains per lane
    fp a, b, c;
    for (int i = 0; i < 10; i++) {
        a.v[i] = io[threadIdx.x * 20 + i];
        b.v[i] = io[threadIdx.x * 20 + 10 + i];
        c.v[i] = a.v[i] ^ 5;
    }
#pragma unroll 1
    for (int it = 0; it < n; it++) {
        a = fp_mul(a, b);
        c = fp_mul(c, b);
    }
    if ((a.v[0] ^ c.v[0]) == 0x7fffffff) io[0] = a.v[1];
}
template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_ecadd(i32 *io, int n) {
    ec_xyzz<FpField> acc;
    fp x, y;
    for (int i = 0; i < 10; i++) {
        x.v[i] = io[threadIdx.x * 20 + i] << 4;
        y.v[i] = io[threadIdx.x * 20 + 10 + i] << 4;
        acc.X.v[i] = io[i] + 1;
        acc.Y.v[i] = io[i] + 2;
        acc.ZZ.v[i] = io[i] + 3;
        acc.ZZZ.v[i] = io[i] + 4;
    }
#pragma unroll 1
    for (int it = 0; it < n; it++) acc = ec_add_affine<FpField>(acc, x, y, it & 1);
    if (acc.X.v[0] == 0x7fffffff) io[0] = acc.Y.v[1];
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_ecadd256(i32 *io, int n) {
    ec_xyzz<FpField> acc;
    fp x, y;
    const int l = threadIdx.x & 63;
    for (int i = 0; i < 10; i++) {
        x.v[i] = io[l * 20 + i] << 4;
        y.v[i] = io[l * 20 + 10 + i] << 4;
        acc.X.v[i] = io[i] + 1;
        acc.Y.v[i] = io[i] + 2;
        acc.ZZ.v[i] = io[i] + 3;
        acc.ZZZ.v[i] = io[i] + 4;
    }
#pragma unroll 1
    for (int it = 0; it < n; it++) acc = ec_add_affine<FpField>(acc, x, y, it & 1);
    if (acc.X.v[0] == 0x7fffffff) io[0] = acc.Y.v[1];
}
__global__ void __launch_bounds__(256) k_mad_i64(uint64_t *out, uint32_t seed, int n) {
    int64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    int32_t b = seed | 1, c = seed * 7 + 3;
    for (int it = 0; it < n; it++) {
        asm volatile("v_mad_i64_i32 %0, s[10:11], %8, %9, %0\n\tv_mad_i64_i32 %1, s[10:11], %8, %9, %1\n\t"
                     "v_mad_i64_i32 %2, s[10:11], %8, %9, %2\n\tv_mad_i64_i32 %3, s[10:11], %8, %9, %3\n\t"
                     "v_mad_i64_i32 %4, s[10:11], %8, %9, %4\n\tv_mad_i64_i32 %5, s[10:11], %8, %9, %5\n\t"
                     "v_mad_i64_i32 %6, s[10:11], %8, %9, %6\n\tv_mad_i64_i32 %7, s[10:11], %8, %9, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c)
                     : "s10", "s11");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// the bucket loop of the MSM itself (msm_bucket_lane: gathers one iteration ahead) on synthetic entries: every lane owns `cnt` random
// point indices (uniform trip count), or cnt * (0.75 .. 1.25) when `spread` (the Poisson spread of real buckets)
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_bucket(const u64 *points, const u32 *entries, u32 cnt, u32 spread, i32 *out) {
    u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    u32 mine = spread ? cnt - cnt / 4 + (lane * 2654435761u >> 16) % (cnt / 2 + 1) : cnt;
    ec_xyzz<FpField> acc = ec_infinity<FpField>();
    msm_bucket_lane<FpField>(acc, points, entries, lane * (cnt + cnt / 4 + 1), 0, mine, 1);
    if (acc.X.v[0] == 0x7fffffff) out[0] = acc.Y.v[1];
}

template <class K>
static double time_ms(K launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, simds = cus * 4;
    printf("device CUs %d clock %d kHz\n", cus, pr.clockRate);
    i32 *io;
    hipMalloc(&io, 64 * 20 * 4 + 4096);
    std::vector<i32> h(64 * 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = (i32)((i * 2654435761u) & 0x1ffffff);
    hipMemcpy(io, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    uint64_t *out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
    {
        const int n = 4096;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mad_i64, dim3(cus * 8), dim3(256), 0, 0, out, 12345u, n); });
        double ops = (double)cus * 8 * 256 * n * 8;
        printf("v_mad_i64_i32 (8 independent chains, 8 waves/SIMD)   %.2f Tlane-ops/s\n", ops / ms / 1e9);
    }
    const int n = 20000;
#define RUN_MUL(W)                                                                                                                  \
    {                                                                                                                                \
        double ms = time_ms([&] { hipLaunchKernelGGL(k_fpmul<W>, dim3(simds * W), dim3(64), 0, 0, io, n); });                        \
        double ms2 = time_ms([&] { hipLaunchKernelGGL(k_fpmul2<W>, dim3(simds * W), dim3(64), 0, 0, io, n); });                      \
        printf("fp_mul chain, %d wave(s)/SIMD: %.1f ns per mul per wave, %.2f Tlane-instr/s | two chains per lane: %.1f ns, %.2f T\n", W, \
               ms * 1e6 / n, FP_MUL_INSTR * 64.0 * simds * W * n / ms / 1e9, ms2 * 1e6 / (2.0 * n),                                  \
               FP_MUL_INSTR * 64.0 * simds * W * 2.0 * n / ms2 / 1e9);                                                               \
    }
    RUN_MUL(1) RUN_MUL(2) RUN_MUL(4) RUN_MUL(8)
    const int na = 2000;
#define RUN_ADD(W)                                                                                                       \
    {                                                                                                                     \
        double ms = time_ms([&] { hipLaunchKernelGGL(k_ecadd<W>, dim3(simds * W), dim3(64), 0, 0, io, na); });            \
        printf("ec_add_affine chain, %d wave(s)/SIMD: %.2f us per addition per wave, %.2f G additions/s\n", W, ms * 1e3 / na, \
               64.0 * simds * W * na / ms / 1e6);                                                                         \
    }
    RUN_ADD(1) RUN_ADD(2) RUN_ADD(3) RUN_ADD(4)
#define RUN_ADD256(W, ROUNDS)                                                                                                       \
    {                                                                                                                                \
        double ms = time_ms([&] { hipLaunchKernelGGL(k_ecadd256<W>, dim3(cus * W * ROUNDS), dim3(256), 0, 0, io, na); });             \
        printf("ec_add_affine chain, 256-thread workgroups, %d wave(s)/SIMD x %d rounds: %.2f G additions/s\n", W, ROUNDS,           \
               256.0 * cus * W * ROUNDS * na / ms / 1e6);                                                                            \
    }
    RUN_ADD256(1, 1) RUN_ADD256(2, 1) RUN_ADD256(2, 4) RUN_ADD256(3, 1)
    {   // one-wave workgroups again, but four rounds of them (the shape of the round-3 bucket kernel)
        double ms = time_ms([&] { hipLaunchKernelGGL(k_ecadd<2>, dim3(simds * 2 * 4), dim3(64), 0, 0, io, na); });
        printf("ec_add_affine chain, 64-thread workgroups, 2 waves/SIMD x 4 rounds: %.2f G additions/s\n", 64.0 * simds * 2 * 4 * na / ms / 1e6);
    }
    {
        // synthetic bucket work: 2^19 lanes (2048 workgroups) x 128 entries over 2^22 points (valid curve points are not needed for
        // timing: the formulas do not branch on the values, except x = 0 which random words avoid)
        const u32 npts = 1u << 22, lanes = 1u << 19, cnt = 128, stride = cnt + cnt / 4 + 1;
        std::vector<u64> hp((size_t)npts * 8);
        u64 x = 88172645463325252ULL;
        for (auto &v : hp) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            v = x & 0x0fffffffffffffffULL;
        }
        std::vector<u32> he((size_t)lanes * stride);
        for (auto &v : he) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            v = (u32)((x >> 20) % npts) << 1;
        }
        u64 *dp;
        u32 *de;
        hipMalloc(&dp, hp.size() * 8);
        hipMalloc(&de, he.size() * 4);
        hipMemcpy(dp, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(de, he.data(), he.size() * 4, hipMemcpyHostToDevice);
#define RUN_BUCKET(W, SPREAD)                                                                                                        \
    {                                                                                                                                \
        double ms = time_ms([&] { hipLaunchKernelGGL(k_bucket<W>, dim3(lanes / 256), dim3(256), 0, 0, dp, de, cnt, SPREAD, io); });   \
        printf("bucket loop with gathers, %d wave(s)/SIMD, %s trip counts: %.2f ms, %.2f G additions/s\n", W,                         \
               SPREAD ? "spread" : "uniform", ms, (double)lanes * cnt / ms / 1e6);                                                    \
    }
        RUN_BUCKET(1, 0) RUN_BUCKET(2, 0) RUN_BUCKET(2, 1) RUN_BUCKET(3, 0)
        // the same with every lane reading ONE point (no gather traffic: L1 hits)
        for (auto &v : he) v = 0;
        hipMemcpy(de, he.data(), he.size() * 4, hipMemcpyHostToDevice);
        printf("same, all entries = point 0:\n");
        RUN_BUCKET(2, 0)
    }
    return 0;
}
