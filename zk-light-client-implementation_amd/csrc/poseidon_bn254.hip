// Poseidon-BN254 permutation, leaf hashing and Merkle tree kernels (gfx950) + C ABI:
// the hasher of the final plonky2 recursion (PoseidonBN128GoldilocksConfig,
// crypto/plonky2_bn128/src/config.rs:21-28,132-199; used at
// near_bft_finality/src/bin/prove_block.rs:279-287).  Digests are the 32-byte little-endian
// REGULAR (non-Montgomery) Fr value = PoseidonBN128HashOut::to_bytes (config.rs:36-46),
// stored as 4 u64 so the tree buffer has the same shape as the Goldilocks one.
#include "poseidon_bn254.cuh"
#include "zklc_internal.h"
#include <stdlib.h>

__global__ void __launch_bounds__(64) poseidon_bn254_permute_kernel(u64 *states, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 *p = reinterpret_cast<u32 *>(states + (size_t)i * 16);
    fr s[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32 w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) w[q] = p[8 * k + q];
        s[k] = fr_from_regular(w);
    }
    poseidon_bn254_permute(s);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32 w[8];
        fr_to_regular(w, s[k]);
#pragma unroll
        for (int q = 0; q < 8; q++) p[8 * k + q] = w[q];
    }
}

__global__ void __launch_bounds__(64)
bn254_hash_leaves_kernel(const u64 *__restrict__ mat, size_t stride, size_t leaf_stride, u32 width, u32 n_leaves,
                         u64 *__restrict__ digests) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    u32 h[8];
    poseidon_bn254_hash_or_noop(mat + (size_t)i * leaf_stride, stride, width, h);
    uint4 *o = reinterpret_cast<uint4 *>(digests + (size_t)i * 4);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

__global__ void __launch_bounds__(64) bn254_merkle_level_kernel(const u64 *__restrict__ children, u64 *__restrict__ parents, u32 n_parents) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents) return;
    const uint4 *c = reinterpret_cast<const uint4 *>(children + (size_t)i * 8);
    uint4 a = c[0], b = c[1], cc = c[2], dd = c[3];
    u32 l[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, r[8] = {cc.x, cc.y, cc.z, cc.w, dd.x, dd.y, dd.z, dd.w}, h[8];
    poseidon_bn254_two_to_one(l, r, h);
    uint4 *o = reinterpret_cast<uint4 *>(parents + (size_t)i * 4);
    o[0] = make_uint4(h[0], h[1], h[2], h[3]);
    o[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// ---- four lanes per permutation (poseidon_bn254_permute_coop): the small trees of the last recursion
#define BN254_COOP_MAX (1u << 14)      // items (leaves / parents) up to which a launch is latency-bound: 2^14 x 4 lanes = one wave per SIMD

__global__ void __launch_bounds__(256)
bn254_hash_leaves_coop_kernel(const u64 *__restrict__ mat, size_t stride, size_t leaf_stride, u32 width, u32 n_leaves,
                              u64 *__restrict__ digests) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 i = t >> 2, q = t & 3;
    bool live = i < n_leaves;
    if (!live) i = n_leaves - 1;       // the quad broadcast needs all four lanes: idle quads redo the last leaf, without storing
    const u64 *in = mat + (size_t)i * leaf_stride;
    if (width <= 3) {                  // hash_or_noop: the elements themselves (config.rs:174-186)
        if (live && q == 0) {
            u32 h[8];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                u64 e = (u32)k < width ? in[(size_t)k * stride] : 0;
                h[2 * k] = (u32)e;
                h[2 * k + 1] = (u32)(e >> 32);
            }
            uint4 *o = reinterpret_cast<uint4 *>(digests + (size_t)i * 4);
            o[0] = make_uint4(h[0], h[1], h[2], h[3]);
            o[1] = make_uint4(h[4], h[5], 0, 0);
        }
        return;
    }
    fr s = fr_zero();
#pragma unroll 1
    for (u32 off = 0; off < width; off += 9) {
        // lane q = 1..3 absorbs elements [off + 3 (q - 1), + 3) into ITS state word (hash_no_pad: three Fr of three elements each)
        u32 o = off + 3 * (q - 1);
        if (q != 0 && o < width) s = pbn_pack3(in + (size_t)o * stride, stride, width - o < 3 ? width - o : 3);
        poseidon_bn254_permute_coop(s, q);
    }
    if (live && q == 0) {
        u32 h[8];
        fr_to_regular(h, s);
        uint4 *o = reinterpret_cast<uint4 *>(digests + (size_t)i * 4);
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[1] = make_uint4(h[4], h[5], h[6], h[7]);
    }
}

__global__ void __launch_bounds__(256)
bn254_merkle_level_coop_kernel(const u64 *__restrict__ children, u64 *__restrict__ parents, u32 n_parents) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 i = t >> 2, q = t & 3;
    bool live = i < n_parents;
    if (!live) i = n_parents - 1;
    // two_to_one(l, r) = permute([0, 0, l, r])[0]: lanes 2 and 3 load one child each
    fr s = fr_zero();
    if (q >= 2) {
        const uint4 *c = reinterpret_cast<const uint4 *>(children + (size_t)i * 8 + (q - 2) * 4);
        uint4 a = c[0], b = c[1];
        u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        s = fr_from_regular(w);
    }
    poseidon_bn254_permute_coop(s, q);
    if (live && q == 0) {
        u32 h[8];
        fr_to_regular(h, s);
        uint4 *o = reinterpret_cast<uint4 *>(parents + (size_t)i * 4);
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[1] = make_uint4(h[4], h[5], h[6], h[7]);
    }
}

extern "C" int32_t zklc_poseidon_bn254_permute_dev(zklc_ctx *ctx, void *stream, uint64_t *d_states, uint32_t n) {
    if (!ctx || (n && !d_states)) return ZKLC_ERR_INVALID_ARG;
    if (n == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(poseidon_bn254_permute_kernel, dim3((n + 63) / 64), dim3(64), 0, zklc_pick_stream(ctx, stream), d_states, n);
    ZKLC_HIP(ctx, hipGetLastError());
    return ZKLC_OK;
}

extern "C" int32_t zklc_poseidon_bn254_permute(zklc_ctx *ctx, uint64_t *states, uint32_t n) {
    if (!ctx || (n && !states)) return ZKLC_ERR_INVALID_ARG;
    if (n == 0) return ZKLC_OK;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t bytes = (size_t)n * 128;
    void *d;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, bytes, &d))) return rc;
    ZKLC_HIP(ctx, hipMemcpyAsync(d, states, bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = zklc_poseidon_bn254_permute_dev(ctx, ctx->stream, (uint64_t *)d, n))) return rc;
    ZKLC_HIP(ctx, zklc_readback_async(states, d, bytes, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}

extern "C" int32_t zklc_bn254_merkle_commit_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_mat, uint64_t stride, uint32_t log_leaves,
                                                uint32_t width, uint32_t cap_height, uint64_t *d_tree) {
    if (!ctx || !d_mat || !d_tree || log_leaves > 30 || cap_height > log_leaves || width == 0) return ZKLC_ERR_INVALID_ARG;
    if (stride < (1ULL << log_leaves)) return ZKLC_ERR_INVALID_ARG;
    return zklc_bn254_merkle_commit_strided(ctx, zklc_pick_stream(ctx, stream), d_mat, stride, 1, log_leaves, width, cap_height, d_tree);
}

int32_t zklc_bn254_merkle_commit_strided(zklc_ctx *ctx, hipStream_t st, const uint64_t *d_mat, uint64_t stride, uint64_t leaf_stride,
                                         uint32_t log_leaves, uint32_t width, uint32_t cap_height, uint64_t *d_tree) {
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    u32 n = 1u << log_leaves;
    // A/B switch: ZKLC_BN254_COOP=<log2 of the largest item count hashed four lanes per permutation> (0: never)
    static const u32 coop_max = [] {
        const char *v = getenv("ZKLC_BN254_COOP");
        if (!v) return BN254_COOP_MAX;
        int k = atoi(v);
        return k <= 0 ? 0u : 1u << (k > 30 ? 30 : k);
    }();
    const bool coop = coop_max != 0;
    if (coop && n <= coop_max)
        hipLaunchKernelGGL(bn254_hash_leaves_coop_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, st, d_mat, (size_t)stride,
                           (size_t)leaf_stride, width, n, d_tree);
    else
        hipLaunchKernelGGL(bn254_hash_leaves_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_mat, (size_t)stride, (size_t)leaf_stride,
                           width, n, d_tree);
    ZKLC_HIP(ctx, hipGetLastError());
    u64 *level = d_tree;
    for (u32 l = 0; l < log_leaves - cap_height; l++) {
        u32 parents = n >> (l + 1);
        u64 *next = level + (4ULL << (log_leaves - l));
        if (coop && parents <= coop_max)
            hipLaunchKernelGGL(bn254_merkle_level_coop_kernel, dim3((4 * parents + 255) / 256), dim3(256), 0, st, (const u64 *)level, next,
                               parents);
        else
            hipLaunchKernelGGL(bn254_merkle_level_kernel, dim3((parents + 63) / 64), dim3(64), 0, st, (const u64 *)level, next, parents);
        ZKLC_HIP(ctx, hipGetLastError());
        level = next;
    }
    return ZKLC_OK;
}

extern "C" int32_t zklc_bn254_merkle_commit(zklc_ctx *ctx, const uint64_t *mat, uint64_t stride, uint32_t log_leaves, uint32_t width,
                                            uint32_t cap_height, uint64_t *tree_out) {
    if (!ctx || !mat || !tree_out || log_leaves > 30 || cap_height > log_leaves || width == 0) return ZKLC_ERR_INVALID_ARG;
    if (stride < (1ULL << log_leaves)) return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    size_t in_bytes = (size_t)stride * width * 8, tree_bytes = zklc_gl_merkle_tree_words(log_leaves, cap_height) * 8;
    void *dm, *dt;
    int32_t rc;
    if ((rc = zklc_stage(ctx, 0, in_bytes, &dm))) return rc;
    if ((rc = zklc_stage(ctx, 1, tree_bytes, &dt))) return rc;
    ZKLC_HIP(ctx, hipMemcpyAsync(dm, mat, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = zklc_bn254_merkle_commit_dev(ctx, ctx->stream, (const uint64_t *)dm, stride, log_leaves, width, cap_height, (uint64_t *)dt)))
        return rc;
    ZKLC_HIP(ctx, zklc_readback_async(tree_out, dt, tree_bytes, ctx->stream));
    ZKLC_HIP(ctx, zklc_stream_wait(ctx->stream));
    return ZKLC_OK;
}
