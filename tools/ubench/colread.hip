// Microbenchmark: how fast can a kernel stream a column-major matrix (W columns x N points, one point per lane or several)?
// The quotient / FRI-combine kernels of the plonky2 prover read ~230-355 columns per point this way.
//   hipcc --offload-arch=gfx950 -O3 -o colread tools/ubench/colread.hip && ./colread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int PTS, int UNROLL>
__global__ void __launch_bounds__(256) colread(const u64 *__restrict__ m, size_t N, int W, u64 *__restrict__ out) {
    size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * PTS;
    if (p >= N) return;
    u64 acc[PTS];
#pragma unroll
    for (int k = 0; k < PTS; k++) acc[k] = 0;
    int j = 0;
    for (; j + UNROLL <= W; j += UNROLL) {
        u64 v[UNROLL][PTS];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const u64 *src = m + (size_t)(j + u) * N + p;
            if (PTS == 1) v[u][0] = src[0];
            else if (PTS == 2) { ulonglong2 t = *(const ulonglong2 *)src; v[u][0] = t.x; v[u][1] = t.y; }
            else { ulonglong2 t0 = ((const ulonglong2 *)src)[0], t1 = ((const ulonglong2 *)src)[1]; v[u][0] = t0.x; v[u][1] = t0.y; v[u][2 % PTS] = t1.x; v[u][3 % PTS] = t1.y; }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int k = 0; k < PTS; k++) acc[k] = acc[k] * 3 + v[u][k];
    }
    for (; j < W; j++)
#pragma unroll
        for (int k = 0; k < PTS; k++) acc[k] = acc[k] * 3 + m[(size_t)j * N + p + k];
#pragma unroll
    for (int k = 0; k < PTS; k++) out[p + k] = acc[k];
}

template <int PTS, int UNROLL> static float run(const u64 *m, size_t N, int W, u64 *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid((unsigned)((N / PTS + 255) / 256));
    hipLaunchKernelGGL((colread<PTS, UNROLL>), grid, dim3(256), 0, 0, m, N, W, out);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((colread<PTS, UNROLL>), grid, dim3(256), 0, 0, m, N, W, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const int W = 234;
    for (int bits = 20; bits <= 21; bits++) {
        size_t N = (size_t)1 << bits;
        u64 *m, *out;
        CHECK(hipMalloc(&m, (size_t)W * N * 8));
        CHECK(hipMalloc(&out, N * 8));
        CHECK(hipMemset(m, 1, (size_t)W * N * 8));
        double gb = (double)W * N * 8 / 1e9;
        float t;
        t = run<1, 4>(m, N, W, out);  printf("N=2^%d  1 pt/lane, 4 loads in flight:  %.3f ms  %.0f GB/s\n", bits, t, gb / t * 1e3);
        t = run<1, 8>(m, N, W, out);  printf("N=2^%d  1 pt/lane, 8 loads in flight:  %.3f ms  %.0f GB/s\n", bits, t, gb / t * 1e3);
        t = run<1, 16>(m, N, W, out); printf("N=2^%d  1 pt/lane, 16 loads in flight: %.3f ms  %.0f GB/s\n", bits, t, gb / t * 1e3);
        t = run<2, 4>(m, N, W, out);  printf("N=2^%d  2 pt/lane (16 B), 4 in flight:  %.3f ms  %.0f GB/s\n", bits, t, gb / t * 1e3);
        t = run<2, 8>(m, N, W, out);  printf("N=2^%d  2 pt/lane (16 B), 8 in flight:  %.3f ms  %.0f GB/s\n", bits, t, gb / t * 1e3);
        t = run<4, 4>(m, N, W, out);  printf("N=2^%d  4 pt/lane (32 B), 4 in flight:  %.3f ms  %.0f GB/s\n", bits, t, gb / t * 1e3);
        hipFree(m); hipFree(out);
    }
    return 0;
}
