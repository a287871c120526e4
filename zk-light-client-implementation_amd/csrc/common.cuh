// Shared qualifiers and small helpers for the zklc gfx950 kernels.
//
// All arithmetic headers (*.cuh) are written as ZKLC_HD inline functions so the
// exact same source can be instantiated inside a HIP kernel (the product) and
// inside tests/hostsim (a g++ build used ONLY by the CPU test-suite to check
// the arithmetic against the oracle without a GPU).  There is no CPU product
// path: the C ABI in zklc_api.hip launches HIP kernels and nothing else.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZKLC_HD __device__ __forceinline__
#define ZKLC_D __device__ __forceinline__
#define ZKLC_M __device__ __forceinline__   /* member functions */
#else
#define ZKLC_HD static inline __attribute__((always_inline))
#define ZKLC_D static inline
#define ZKLC_M inline
#endif

typedef uint32_t u32;
typedef uint64_t u64;

// (c2:c1:c0) += a*b   -- 96-bit column accumulator; lowers to one
// v_mad_u64_u32 (with carry-out) + one v_addc_co_u32 on gfx950.
ZKLC_HD void mac96(u64 &lo, u32 &hi, u32 a, u32 b) {
    u64 p = (u64)a * b;
    lo += p;
    hi += (lo < p);
}
