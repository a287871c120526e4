/* zklc.h -- C ABI of libzklc_mi355.so: MI355X (gfx950) kernels for the NEAR
 * zk-light-client signature-aggregation hot path.
 *
 * The reference (ZpokenWeb3/zk-light-client-implementation) has no FFI for this
 * path; each entry point names the reference interface it replaces
 * (paths relative to the reference root) and INTEGRATION.md shows the Rust
 * `extern "C"` / cgo binding a maintainer would add at that call site.
 *
 * Conventions
 *   - all integers little-endian; status = int32_t, 0 = ZKLC_OK, negative = error
 *   - `*_dev` entry points take DEVICE pointers plus a hipStream_t (as void*;
 *     NULL = HIP's legacy default stream, as everywhere in HIP) and only
 *     enqueue work on that stream;
 *     the plain entry points take caller-owned HOST pointers, stage through the
 *     context's device buffers and return after the result is on the host.
 *   - never aborts, never unwinds across the boundary; re-entrant per context
 *     (one context per host thread / per process-per-GPU rank).
 */
#ifndef ZKLC_H
#define ZKLC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct zklc_ctx zklc_ctx;

#define ZKLC_OK 0
#define ZKLC_ERR_INVALID_ARG (-1)
#define ZKLC_ERR_OOM (-2)
#define ZKLC_ERR_HIP (-3)
#define ZKLC_ERR_NO_DEVICE (-4)

/* One context per process-per-GPU rank.  Owns a stream, staging buffers and
 * the constant tables (Ed25519 base-point table, NTT twiddles, ...). */
int32_t zklc_init(zklc_ctx **out, int32_t device_id);
void zklc_destroy(zklc_ctx *ctx);
const char *zklc_strerror(int32_t code);
/* last HIP error string seen by this context (for ZKLC_ERR_HIP) */
const char *zklc_last_hip_error(zklc_ctx *ctx);
int32_t zklc_synchronize(zklc_ctx *ctx);
uint32_t zklc_abi_version(void);

/* ---- (a) batched Ed25519 -------------------------------------------------
 * Replaces the per-approval native pre-check loop
 *   near_bft_finality/src/prove_block_data/signatures.rs:70-123 (`sig.verify(msg,&pk)` :79)
 * and the in-tree restatement crypto/plonky2_ed25519/src/curve/eddsa.rs:33-58.
 * pks: n*32 bytes, sigs: n*64 bytes (R || s).  Message: msg_stride == 0 ->
 * one msg_len-byte message shared by all n signatures (NEAR: every validator
 * signs the same Approval bytes, signatures.rs:24-39); otherwise signature i
 * uses msgs + i*msg_stride.  ok[i] = 1 valid / 0 invalid (ed25519-dalek
 * non-strict semantics: s >= l rejected, undecodable A rejected,
 * compress([s]B - [h]A) == R bytes). */
int32_t zklc_ed25519_verify_batch(zklc_ctx *ctx, const uint8_t *pks, const uint8_t *sigs, const uint8_t *msgs,
                                  uint32_t msg_len, uint32_t msg_stride, uint32_t n, uint8_t *ok);
int32_t zklc_ed25519_verify_batch_dev(zklc_ctx *ctx, void *stream, const uint8_t *d_pks, const uint8_t *d_sigs,
                                      const uint8_t *d_msgs, uint32_t msg_len, uint32_t msg_stride, uint32_t n,
                                      uint8_t *d_ok);

/* SHA-512 of n equal-length messages (in + i*stride, len bytes) -> out + 64*i.
 * Replaces sha2::Sha512 at crypto/plonky2_ed25519/src/curve/eddsa.rs:40-42 and
 * the witness side of crypto/plonky2_sha512/src/circuit.rs:308-435. */
int32_t zklc_sha512_batch(zklc_ctx *ctx, const uint8_t *in, uint32_t stride, uint32_t len, uint32_t n, uint8_t *out);
int32_t zklc_sha512_batch_dev(zklc_ctx *ctx, void *stream, const uint8_t *d_in, uint32_t stride, uint32_t len,
                              uint32_t n, uint8_t *d_out);

/* ---- (b) Goldilocks: NTT / LDE / Poseidon / Merkle --------------------------
 * Replace the prover internals of the un-vendored plonky2 fork
 * (wormhole-foundation/plonky2-near@2244a9d, Cargo.toml:44-47) reached from
 * `circuit_data.prove(pw)` at near_bft_finality/src/prove_crypto/ed25519.rs:60,100
 * and recursion.rs:95: `fft`/`ifft`/`coset_fft` (plonky2_field::fft),
 * `PolynomialBatch::from_coeffs` (LDE + Merkle) and `MerkleTree::new`.
 * Elements are canonical u64 < p = 2^64 - 2^32 + 1.  A batch is POLY-MAJOR:
 * polynomial b, index i at data[b * 2^log_n + i].
 * values[k] = sum_j coeffs[j] w^(jk), w = 1753635133440165772^(2^32 / n)
 * (gnark-plonky2-verifier/goldilocks/base.go:33-42). */
#define ZKLC_NTT_INVERSE 1u     /* inverse transform (includes the 1/n factor) */
#define ZKLC_NTT_IN_BITREV 2u   /* input is in bit-reversed order (output natural) */
#define ZKLC_NTT_OUT_BITREV 4u  /* leave the output in bit-reversed order (skips the permutation pass) */
/* In-place batched NTT.  coset_shift != 0 (forward, natural-order input only):
 * coefficient j is first multiplied by coset_shift^j (= evaluation on shift*<w>). */
int32_t zklc_gl_ntt(zklc_ctx *ctx, uint64_t *data, uint32_t log_n, uint32_t batch, uint32_t flags, uint64_t coset_shift);
int32_t zklc_gl_ntt_dev(zklc_ctx *ctx, void *stream, uint64_t *d_data, uint32_t log_n, uint32_t batch, uint32_t flags,
                        uint64_t coset_shift);
/* Low-degree extension: coeffs (batch x 2^log_n, natural order) -> evaluations on
 * coset_shift * <w_N>, N = 2^(log_n + rate_bits), out = batch x N; flags: ZKLC_NTT_OUT_BITREV
 * gives the order plonky2 commits to (leaf i = evaluation at shift * w^bitrev(i)). */
int32_t zklc_gl_lde(zklc_ctx *ctx, const uint64_t *coeffs, uint32_t log_n, uint32_t rate_bits, uint32_t batch,
                    uint64_t coset_shift, uint64_t *out, uint32_t flags);
int32_t zklc_gl_lde_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_coeffs, uint32_t log_n, uint32_t rate_bits,
                        uint32_t batch, uint64_t coset_shift, uint64_t *d_out, uint32_t flags);
/* Poseidon-Goldilocks permutation of n states of 12 elements, in place
 * (gnark-plonky2-verifier/poseidon/goldilocks.go:30-37). */
int32_t zklc_poseidon_gl_permute(zklc_ctx *ctx, uint64_t *states, uint32_t n);
int32_t zklc_poseidon_gl_permute_dev(zklc_ctx *ctx, void *stream, uint64_t *d_states, uint32_t n);
/* Merkle tree with cap (plonky2 `MerkleTree::new`; verification side restated in
 * gnark-plonky2-verifier/fri/fri.go:97-144).  Leaf i = (mat[p * stride + i])_{p < width},
 * leaf digest = hash_or_noop, inner node = two_to_one.  tree receives all digest levels,
 * leaves first: level l (2^(log_leaves - l) digests of 4 u64) at word offset
 * sum_{j<l} 4 * 2^(log_leaves - j); the last level (l = log_leaves - cap_height) is the cap.
 * zklc_gl_merkle_tree_words gives the total size in u64 words. */
uint64_t zklc_gl_merkle_tree_words(uint32_t log_leaves, uint32_t cap_height);
int32_t zklc_gl_merkle_commit(zklc_ctx *ctx, const uint64_t *mat, uint64_t stride, uint32_t log_leaves, uint32_t width,
                              uint32_t cap_height, uint64_t *tree_out);
int32_t zklc_gl_merkle_commit_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_mat, uint64_t stride, uint32_t log_leaves,
                                  uint32_t width, uint32_t cap_height, uint64_t *d_tree);

/* Poseidon-BN254 (iden3 t = 4) hasher of the last recursion -- replaces
 * crypto/plonky2_bn128/src/poseidon_bn128.rs:18-108 (`permution`) and the `Hasher` impl
 * crypto/plonky2_bn128/src/config.rs:132-199 (hash_no_pad / hash_or_noop / two_to_one).
 * states: n x 4 Fr, each 4 little-endian u64 in REGULAR (non-Montgomery) canonical form
 * (= ff's `to_repr`).  The Merkle entry points take the same poly-major Goldilocks matrix as
 * zklc_gl_merkle_commit and produce a tree of the same shape whose digests are the 32-byte
 * little-endian Fr value (= PoseidonBN128HashOut::to_bytes). */
int32_t zklc_poseidon_bn254_permute(zklc_ctx *ctx, uint64_t *states, uint32_t n);
int32_t zklc_poseidon_bn254_permute_dev(zklc_ctx *ctx, void *stream, uint64_t *d_states, uint32_t n);
int32_t zklc_bn254_merkle_commit(zklc_ctx *ctx, const uint64_t *mat, uint64_t stride, uint32_t log_leaves, uint32_t width,
                                 uint32_t cap_height, uint64_t *tree_out);
int32_t zklc_bn254_merkle_commit_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_mat, uint64_t stride, uint32_t log_leaves,
                                     uint32_t width, uint32_t cap_height, uint64_t *d_tree);

/* ---- (c) BN254 ---------------------------------------------------------------
 * G1 multi-scalar multiplication sum_i scalars[i] * points[i].
 * Replaces gnark-crypto `bn254.G1Affine.MultiExp` (un-vendored; gnark-plonky2-verifier/go.mod:9)
 * called from `groth16.Prove` at gnark-plonky2-verifier/cmd/web-api.go:77.
 * points: n affine points in gnark-crypto's memory layout = x then y, each 4 little-endian
 * u64 limbs in Montgomery form (x * 2^256 mod p); (0, 0) is the point at infinity.
 * scalars: n x 4 little-endian u64, REGULAR (non-Montgomery) form, reduced (< r).
 * out_affine: 8 u64 in the same layout, canonical; *out_is_infinity = 1 when the sum is
 * the point at infinity (then out_affine is zero).  Pointers must be 16-byte aligned. */
int32_t zklc_bn254_g1_msm(zklc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, uint64_t n, uint64_t *out_affine,
                          uint32_t *out_is_infinity);
uint64_t zklc_bn254_g1_msm_workspace_bytes(uint64_t n);
int32_t zklc_bn254_g1_msm_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, const uint64_t *d_scalars, uint64_t n,
                              uint64_t *d_out_affine, uint32_t *d_out_is_infinity, void *d_workspace, uint64_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif
