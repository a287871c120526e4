// gfx950 VALU micro-benchmark: issue rate of the integer / fp64 instructions the
// big-integer kernels (fe25519, Goldilocks, BN254) are built from.
// Build: hipcc --offload-arch=gfx950 -O3 valu_ubench.hip -o valu_ubench ; run on the GPU box.
// Prints one line per instruction: Tops/s over the whole chip and cycles per
// wave-instruction per SIMD (at the measured clock estimate from v_fma_f32 = 2 cyc).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define ITERS 4096
#define REP8(x) x x x x x x x x

// 8 independent chains per lane; each asm body is one instruction on chain registers
#define DEFINE_KERNEL(NAME, DECL, BODY, SINK)                                          \
    __global__ void __launch_bounds__(256) NAME(uint64_t *out, uint32_t seed) {         \
        DECL;                                                                          \
        for (int it = 0; it < ITERS; it++) {                                           \
            BODY BODY BODY BODY                                                        \
        }                                                                              \
        SINK;                                                                          \
    }

#define DECL32                                                                                                   \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4,    \
             a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7, b = seed | 1, c = seed * 7 + 3
#define SINK32 out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7
#define DECL64                                                                                                   \
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4,    \
             a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;                                            \
    uint32_t b = seed | 1, c = seed * 7 + 3;                                                                    \
    uint64_t d = ((uint64_t)seed << 20) | 5
#define SINK64 out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7

#define OP8_32(INS)                                                                     \
    asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t" INS \
                     " %4, %4, %8\n\t" INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8"        \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                 : "v"(b));
#define OP8_32_3(INS)                                                                                            \
    asm volatile(INS " %0, %0, %8, %9\n\t" INS " %1, %1, %8, %9\n\t" INS " %2, %2, %8, %9\n\t" INS             \
                     " %3, %3, %8, %9\n\t" INS " %4, %4, %8, %9\n\t" INS " %5, %5, %8, %9\n\t" INS             \
                     " %6, %6, %8, %9\n\t" INS " %7, %7, %8, %9"                                               \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                 : "v"(b), "v"(c));

DEFINE_KERNEL(k_mul_lo_u32, DECL32, OP8_32("v_mul_lo_u32"), SINK32)
DEFINE_KERNEL(k_mul_hi_u32, DECL32, OP8_32("v_mul_hi_u32"), SINK32)
DEFINE_KERNEL(k_mul_u32_u24, DECL32, OP8_32("v_mul_u32_u24"), SINK32)
DEFINE_KERNEL(k_mul_hi_u32_u24, DECL32, OP8_32("v_mul_hi_u32_u24"), SINK32)
DEFINE_KERNEL(k_add_u32, DECL32, OP8_32("v_add_u32"), SINK32)
DEFINE_KERNEL(k_xor_b32, DECL32, OP8_32("v_xor_b32"), SINK32)
DEFINE_KERNEL(k_mad_u32_u24, DECL32, OP8_32_3("v_mad_u32_u24"), SINK32)
DEFINE_KERNEL(k_add3_u32, DECL32, OP8_32_3("v_add3_u32"), SINK32)
DEFINE_KERNEL(k_alignbit, DECL32, OP8_32_3("v_alignbit_b32"), SINK32)
DEFINE_KERNEL(k_lshl_add_u32, DECL32, asm volatile("v_lshl_add_u32 %0, %0, 3, %8\n\tv_lshl_add_u32 %1, %1, 3, %8\n\tv_lshl_add_u32 %2, %2, 3, %8\n\tv_lshl_add_u32 %3, %3, 3, %8\n\tv_lshl_add_u32 %4, %4, 3, %8\n\tv_lshl_add_u32 %5, %5, 3, %8\n\tv_lshl_add_u32 %6, %6, 3, %8\n\tv_lshl_add_u32 %7, %7, 3, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));, SINK32)
DEFINE_KERNEL(k_fma_f32, DECL32, OP8_32_3("v_fma_f32"), SINK32)

#define OP8_MAD64                                                                                                 \
    asm volatile("v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n\tv_mad_u64_u32 %1, s[10:11], %8, %9, %1\n\t"           \
                 "v_mad_u64_u32 %2, s[10:11], %8, %9, %2\n\tv_mad_u64_u32 %3, s[10:11], %8, %9, %3\n\t"           \
                 "v_mad_u64_u32 %4, s[10:11], %8, %9, %4\n\tv_mad_u64_u32 %5, s[10:11], %8, %9, %5\n\t"           \
                 "v_mad_u64_u32 %6, s[10:11], %8, %9, %6\n\tv_mad_u64_u32 %7, s[10:11], %8, %9, %7"               \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                 : "v"(b), "v"(c)                                                                                 \
                 : "s10", "s11");
DEFINE_KERNEL(k_mad_u64_u32, DECL64, OP8_MAD64, SINK64)

#define OP8_LSHLADD64                                                                                             \
    asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n\tv_lshl_add_u64 %1, %1, 0, %8\n\tv_lshl_add_u64 %2, %2, 0, %8\n\t" \
                 "v_lshl_add_u64 %3, %3, 0, %8\n\tv_lshl_add_u64 %4, %4, 0, %8\n\tv_lshl_add_u64 %5, %5, 0, %8\n\t" \
                 "v_lshl_add_u64 %6, %6, 0, %8\n\tv_lshl_add_u64 %7, %7, 0, %8"                                   \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                 : "v"(d));
DEFINE_KERNEL(k_lshl_add_u64, DECL64, OP8_LSHLADD64, SINK64)

#define OP8_LSHR64                                                                                                \
    asm volatile("v_lshrrev_b64 %0, 1, %0\n\tv_lshrrev_b64 %1, 1, %1\n\tv_lshrrev_b64 %2, 1, %2\n\t"             \
                 "v_lshrrev_b64 %3, 1, %3\n\tv_lshrrev_b64 %4, 1, %4\n\tv_lshrrev_b64 %5, 1, %5\n\t"             \
                 "v_lshrrev_b64 %6, 1, %6\n\tv_lshrrev_b64 %7, 1, %7"                                             \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
DEFINE_KERNEL(k_lshrrev_b64, DECL64, OP8_LSHR64, SINK64)

// 64-bit add as add_co + addc pairs (VCC chain, hazard wait states included by hand: s_nop 1)
#define OP8_ADDC                                                                                                  \
    asm volatile("v_add_co_u32 %0, vcc, %0, %8\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"             \
                 "v_add_co_u32 %2, vcc, %2, %8\n\ts_nop 1\n\tv_addc_co_u32 %3, vcc, %3, %9, vcc\n\t"             \
                 "v_add_co_u32 %4, vcc, %4, %8\n\ts_nop 1\n\tv_addc_co_u32 %5, vcc, %5, %9, vcc\n\t"             \
                 "v_add_co_u32 %6, vcc, %6, %8\n\ts_nop 1\n\tv_addc_co_u32 %7, vcc, %7, %9, vcc"                 \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                 : "v"(b), "v"(c)                                                                                 \
                 : "vcc");
DEFINE_KERNEL(k_addco_nop_addc_pairs, DECL32, OP8_ADDC, SINK32)

// carry-flag forms used by the hand-scheduled Poseidon statements (tools/gen_poseidon_asm.py): VOP3 with an SGPR-pair carry
// against the VOP2 + VCC encodings
#define OP8_RAW(TXT, CLOB...)                                                                                     \
    asm volatile(TXT(0) TXT(1) TXT(2) TXT(3) TXT(4) TXT(5) TXT(6) TXT(7)                                           \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                 : "v"(b), "v"(c)                                                                                 \
                 : CLOB);
#define T_ADDCO_E64(i) "v_add_co_u32_e64 %" #i ", s[10:11], %" #i ", %8\n\t"
#define T_ADDCO_E32(i) "v_add_co_u32_e32 %" #i ", vcc, %" #i ", %8\n\t"
#define T_ADDC_E64(i) "v_addc_co_u32_e64 %" #i ", s[10:11], %" #i ", %8, s[12:13]\n\t"
#define T_SUBB_E64(i) "v_subb_co_u32_e64 %" #i ", s[10:11], %" #i ", 0, s[12:13]\n\t"
#define T_CND_E64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[12:13]\n\t"
#define T_CND_E32(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n\t"
#define T_MOV_E32(i) "v_mov_b32_e32 %" #i ", %8\n\t"
#define T_MAD_CARRY(i) "v_mad_u64_u32 %" #i ", s[10:11], %8, %9, %" #i "\n\t"
DEFINE_KERNEL(k_addco_e64, DECL32, OP8_RAW(T_ADDCO_E64, "s10", "s11"), SINK32)
DEFINE_KERNEL(k_addco_e32, DECL32, OP8_RAW(T_ADDCO_E32, "vcc"), SINK32)
DEFINE_KERNEL(k_addc_e64, DECL32, OP8_RAW(T_ADDC_E64, "s10", "s11"), SINK32)
DEFINE_KERNEL(k_subb_e64, DECL32, OP8_RAW(T_SUBB_E64, "s10", "s11"), SINK32)
DEFINE_KERNEL(k_cnd_e64, DECL32, OP8_RAW(T_CND_E64, "s10"), SINK32)
DEFINE_KERNEL(k_cnd_e32, DECL32, OP8_RAW(T_CND_E32, "s10"), SINK32)
DEFINE_KERNEL(k_mov_e32, DECL32, OP8_RAW(T_MOV_E32, "s10"), SINK32)

#define DECLF64                                                                                                  \
    double a0 = 1.0 + threadIdx.x * 1e-9, a1 = a0 + 1e-3, a2 = a0 + 2e-3, a3 = a0 + 3e-3, a4 = a0 + 4e-3,     \
           a5 = a0 + 5e-3, a6 = a0 + 6e-3, a7 = a0 + 7e-3, b = 1.0 + seed * 1e-12, c = seed * 1e-13
#define SINKF64 out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
#define OP8_F64(INS)                                                                                             \
    asm volatile(INS " %0, %0, %8, %9\n\t" INS " %1, %1, %8, %9\n\t" INS " %2, %2, %8, %9\n\t" INS             \
                     " %3, %3, %8, %9\n\t" INS " %4, %4, %8, %9\n\t" INS " %5, %5, %8, %9\n\t" INS             \
                     " %6, %6, %8, %9\n\t" INS " %7, %7, %8, %9"                                               \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                 : "v"(b), "v"(c));
DEFINE_KERNEL(k_fma_f64, DECLF64, OP8_F64("v_fma_f64"), SINKF64)
#define OP8_F64_2(INS)                                                                                           \
    asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t" INS \
                     " %4, %4, %8\n\t" INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8"        \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)           \
                 : "v"(b));
DEFINE_KERNEL(k_mul_f64, DECLF64, OP8_F64_2("v_mul_f64"), SINKF64)
DEFINE_KERNEL(k_add_f64, DECLF64, OP8_F64_2("v_add_f64"), SINKF64)

// dependent chain latency of v_mad_u64_u32 (one chain)
__global__ void __launch_bounds__(64) k_mad_u64_u32_dep(uint64_t *out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x;
    uint32_t b = seed | 1, c = seed * 7 + 3;
    for (int it = 0; it < ITERS; it++) {
        asm volatile(REP8("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n\t") REP8("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n\t")
                     REP8("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n\t") REP8("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n\t")
                     : "+v"(a0) : "v"(b), "v"(c) : "s10", "s11");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}

struct Case {
    const char *name;
    void (*fn)(uint64_t *, uint32_t);
    int ops_per_iter;  // per lane per loop iteration
};

// shader clock in MHz read from the hwmon `freq1_input` of the busiest card (Hz), 0 when sysfs has none
#include <dirent.h>
#include <string>
static double read_sclk_mhz() {
    double best = 0;
    for (int card = 0; card < 64; card++) {
        std::string base = "/sys/class/drm/card" + std::to_string(card) + "/device/hwmon";
        DIR *d = opendir(base.c_str());
        if (!d) continue;
        while (dirent *e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            std::string f = base + "/" + e->d_name + "/freq1_input";
            if (FILE *fp = fopen(f.c_str(), "r")) {
                double hz = 0;
                if (fscanf(fp, "%lf", &hz) == 1 && hz * 1e-6 > best) best = hz * 1e-6;
                fclose(fp);
            }
        }
        closedir(d);
    }
    return best;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    uint64_t *out;
    const int blocks = p.multiProcessorCount * 8, threads = 256;  // 8 blocks x 4 waves = 32 waves/CU = 8 per SIMD
    hipMalloc(&out, (size_t)blocks * threads * 8);
    std::vector<Case> cases = {
        {"v_fma_f32", k_fma_f32, 32},          {"v_add_u32", k_add_u32, 32},
        {"v_xor_b32", k_xor_b32, 32},          {"v_add3_u32", k_add3_u32, 32},
        {"v_lshl_add_u32", k_lshl_add_u32, 32}, {"v_alignbit_b32", k_alignbit, 32},
        {"v_mul_u32_u24", k_mul_u32_u24, 32},  {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 32},
        {"v_mad_u32_u24", k_mad_u32_u24, 32},  {"v_mul_lo_u32", k_mul_lo_u32, 32},
        {"v_mul_hi_u32", k_mul_hi_u32, 32},    {"v_mad_u64_u32", k_mad_u64_u32, 32},
        {"v_lshl_add_u64", k_lshl_add_u64, 32}, {"v_lshrrev_b64", k_lshrrev_b64, 32},
        {"add_co;s_nop1;addc (per pair)", k_addco_nop_addc_pairs, 16},
        {"v_add_co_u32_e64 (sgpr carry out)", k_addco_e64, 32}, {"v_add_co_u32_e32 (vcc)", k_addco_e32, 32},
        {"v_addc_co_u32_e64 (sgpr in/out)", k_addc_e64, 32}, {"v_subb_co_u32_e64", k_subb_e64, 32},
        {"v_cndmask_b32_e64 (sgpr)", k_cnd_e64, 32}, {"v_cndmask_b32_e32 (vcc)", k_cnd_e32, 32},
        {"v_mov_b32_e32", k_mov_e32, 32},
        {"v_fma_f64", k_fma_f64, 32},          {"v_mul_f64", k_mul_f64, 32},
        {"v_add_f64", k_add_f64, 32},
        {"v_fma_f32 (again, last)", k_fma_f32, 32},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // WARM-UP (round 5): ~0.5 s of work before anything is timed, so that the first case does not run on an idle clock (the round-2
    // table had v_fma_f32 first and cold: 52 T, below the v_mov measured later in the same run)
    for (int r = 0; r < 200; r++) hipLaunchKernelGGL(k_fma_f32, dim3(blocks), dim3(threads), 0, 0, out, 7u + r);
    hipDeviceSynchronize();
    printf("sclk after warm-up: %.0f MHz (hwmon freq1_input)\n", read_sclk_mhz());
    const double simds = (double)p.multiProcessorCount * 4;
    const int REPS = 24;
    for (auto &c : cases) {
        for (int r = 0; r < 4; r++) hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(threads), 0, 0, out, 12345u);  // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < REPS / 2; r++) hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(threads), 0, 0, out, 12345u + r);
        hipStreamSynchronize(0);
        double mhz = read_sclk_mhz();                       // sampled in the middle of the timed launches' window
        for (int r = REPS / 2; r < REPS; r++) hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(threads), 0, 0, out, 12345u + r);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double ops = (double)REPS * blocks * threads * ITERS * c.ops_per_iter;
        double rate = ops / (ms * 1e-3);
        // cycles per wave-instruction per SIMD from the SAMPLED clock: sclk * SIMDs / (wave-instructions per second)
        double cyc = mhz > 0 ? mhz * 1e6 * simds / (rate / 64.0) : 0;
        printf("%-34s %8.2f Tlane-ops/s   sclk %5.0f MHz   %6.2f cyc/wave-instr/SIMD\n", c.name, rate / 1e12, mhz, cyc);
    }
    {
        // latency: 1 wave per SIMD
        int b2 = p.multiProcessorCount * 4;
        hipLaunchKernelGGL(k_mad_u64_u32_dep, dim3(b2), dim3(64), 0, 0, out, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mad_u64_u32_dep, dim3(b2), dim3(64), 0, 0, out, 2u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double ns_per = ms * 1e6 / ((double)ITERS * 32);
        printf("v_mad_u64_u32 dependent chain: %.2f ns per instruction (1 wave/SIMD), sclk %.0f MHz\n", ns_per, read_sclk_mhz());
    }
    return 0;
}
