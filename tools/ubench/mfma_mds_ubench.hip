// gfx950 micro-benchmark behind the "MFMA for the Poseidon MDS layer?" decision.
// The MDS layer of 64 states (one wavefront, state-per-lane) as byte-limb GEMMs would take 32 v_mfma_i32_16x16x64_i8
// (12 x 12 six-bit matrix = 12 of the 16 rows and 12 of the 64 K slots of a tile; 16 columns = 16 states x ONE of the 8 byte
// limbs per instruction) plus the VALU work of slicing 64-bit words into bytes and re-assembling 8 partial sums per word.
// The pure VALU form is 375 instructions per wavefront (tools/gen_poseidon_asm.py: 288 v_mad_u64_u32 + reductions).
// This program measures, per SIMD:  (a) cycles per v_mfma_i32_16x16x64_i8 with independent accumulators,
// (b) cycles per v_mad_u64_u32, (c) both interleaved 1 : 8 -- does the matrix pipe run beside the VALU or take its issue slots?
// Build: hipcc --offload-arch=gfx950 -O3 mfma_mds_ubench.hip -o mfma_mds_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 2048
typedef int v4i __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#define MAD(x) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(x) : "v"(m0), "v"(m1) : "s10", "s11");

__global__ void __launch_bounds__(256) k_mfma(uint64_t *out, uint32_t seed) {
    v4i a = {(int)seed, 1, 2, 3}, b = {(int)threadIdx.x, 5, 6, 7}, c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < ITERS; it++) {
        MFMA(c0) MFMA(c1) MFMA(c2) MFMA(c3)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0.x ^ c1.y ^ c2.z ^ c3.w;
}
__global__ void __launch_bounds__(256) k_mad(uint64_t *out, uint32_t seed) {
    uint64_t x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m0 = seed | 1, m1 = threadIdx.x | 3;
    for (int it = 0; it < ITERS; it++) {
        MAD(x0) MAD(x1) MAD(x2) MAD(x3) MAD(x4) MAD(x5) MAD(x6) MAD(x7)
        MAD(x0) MAD(x1) MAD(x2) MAD(x3) MAD(x4) MAD(x5) MAD(x6) MAD(x7)
        MAD(x0) MAD(x1) MAD(x2) MAD(x3) MAD(x4) MAD(x5) MAD(x6) MAD(x7)
        MAD(x0) MAD(x1) MAD(x2) MAD(x3) MAD(x4) MAD(x5) MAD(x6) MAD(x7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void __launch_bounds__(256) k_both(uint64_t *out, uint32_t seed) {
    v4i a = {(int)seed, 1, 2, 3}, b = {(int)threadIdx.x, 5, 6, 7}, c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    uint64_t x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
    uint32_t m0 = seed | 1, m1 = threadIdx.x | 3;
    for (int it = 0; it < ITERS; it++) {
        MFMA(c0) MAD(x0) MAD(x1) MAD(x2) MAD(x3) MAD(x4) MAD(x5) MAD(x6) MAD(x7)
        MFMA(c1) MAD(x0) MAD(x1) MAD(x2) MAD(x3) MAD(x4) MAD(x5) MAD(x6) MAD(x7)
        MFMA(c2) MAD(x0) MAD(x1) MAD(x2) MAD(x3) MAD(x4) MAD(x5) MAD(x6) MAD(x7)
        MFMA(c3) MAD(x0) MAD(x1) MAD(x2) MAD(x3) MAD(x4) MAD(x5) MAD(x6) MAD(x7)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0.x ^ c1.y ^ c2.z ^ c3.w ^ x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

static float run(void (*fn)(uint64_t *, uint32_t), uint64_t *out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 11u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8;      // 8 waves per SIMD
    uint64_t *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    const double clk = p.clockRate * 1e3, simds = p.multiProcessorCount * 4.0, waves_per_simd = 8;
    float t_mfma = run(k_mfma, out, blocks), t_mad = run(k_mad, out, blocks), t_both = run(k_both, out, blocks);
    double cyc_mfma = t_mfma * 1e-3 * clk / (ITERS * 4.0 * waves_per_simd);
    double cyc_mad = t_mad * 1e-3 * clk / (ITERS * 32.0 * waves_per_simd);
    double cyc_both = t_both * 1e-3 * clk / (ITERS * 4.0 * waves_per_simd);   // per group of 1 MFMA + 8 MAD
    printf("device %s, %d CUs, %.0f MHz (nominal)\n", p.name, p.multiProcessorCount, clk / 1e6);
    printf("v_mfma_i32_16x16x64_i8 : %.2f cycles per wave-instruction per SIMD  (%.0f dense int8 TOPS)\n", cyc_mfma,
           32768.0 * simds * clk / cyc_mfma / 1e12);
    printf("v_mad_u64_u32          : %.2f cycles per wave-instruction per SIMD\n", cyc_mad);
    printf("1 MFMA + 8 MAD         : %.2f cycles per group (sum of the parts %.2f, max of the parts %.2f)\n", cyc_both,
           cyc_mfma + 8 * cyc_mad, cyc_mfma > 8 * cyc_mad ? cyc_mfma : 8 * cyc_mad);
    printf("MDS layer of one wavefront (64 states): VALU form 375 instructions = %.0f cycles; MFMA form 32 MFMA = %.0f cycles of the matrix\n"
           "pipe + the byte slicing / re-assembly on the VALU (>= 12 words x (8 + 10) instructions = 216 -> %.0f cycles)\n",
           375 * cyc_mad, 32 * cyc_mfma, 216 * cyc_mad);
    (void)simds;
    return 0;
}
