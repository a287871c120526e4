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

#ifdef __cplusplus
}
#endif
#endif
