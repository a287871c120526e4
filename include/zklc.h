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
#define ZKLC_ERR_IO (-5)        /* a container file could not be opened / written (errno is the caller's to read) */
#define ZKLC_ERR_FORMAT (-6)    /* not a container, another version, truncated, a failed checksum or sections that contradict each other */
#define ZKLC_ERR_NOT_FOUND (-7) /* the container has no section with that tag */

/* One context per process-per-GPU rank.  Owns a stream, staging buffers and
 * the constant tables (Ed25519 base-point table, NTT twiddles, ...). */
int32_t zklc_init(zklc_ctx **out, int32_t device_id);
/* same, with the context's stream created at the device's highest stream priority when `high_priority` != 0: for the short,
 * latency-critical proofs of a dependent chain (the fold of signatures.rs:97-105) sharing the GPU with long throughput work */
int32_t zklc_init_priority(zklc_ctx **out, int32_t device_id, int32_t high_priority);
void zklc_destroy(zklc_ctx *ctx);
const char *zklc_strerror(int32_t code);
/* last HIP error string seen by this context (for ZKLC_ERR_HIP) */
const char *zklc_last_hip_error(zklc_ctx *ctx);
int32_t zklc_synchronize(zklc_ctx *ctx);
/* the context's own non-blocking hipStream_t (what the host-pointer entry points run on); pass it to the *_dev entry
 * points to keep several contexts' work concurrent on one GPU */
void *zklc_stream(zklc_ctx *ctx);
uint32_t zklc_abi_version(void);
/* Device memory for hosts that bind nothing but this ABI (a C / Rust / Go caller need not link the HIP runtime to hold the
 * buffers the *_dev entry points take): alloc returns a ZERO-FILLED buffer on the context's GPU (what
 * zklc_plonky2_witness_run_dev expects of d_wires); copy moves bytes between host and device on the context's stream and returns
 * when they have arrived (to_host != 0: device -> host). */
int32_t zklc_device_alloc(zklc_ctx *ctx, uint64_t bytes, void **d_out);
int32_t zklc_device_free(zklc_ctx *ctx, void *d_ptr);
int32_t zklc_device_copy(zklc_ctx *ctx, void *dst, const void *src, uint64_t bytes, int32_t to_host);

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

/* ---- (b') plonky2 prover -------------------------------------------------------
 * Replaces `CircuitData::prove` of the un-vendored plonky2 fork (plonky2-near@2244a9d `plonk/prover.rs`
 * prove_with_partition_witness: wires commit, Z / partial products, quotient, openings, FRI), reached from
 *   near_bft_finality/src/prove_crypto/ed25519.rs:60,100 (ed25519 proofs, 2^17 rows x 234 wires)
 *   near_bft_finality/src/prove_crypto/recursion.rs:95   (recursion proofs, 2^12 rows x 135 wires)
 *   near_bft_finality/src/bin/prove_block.rs:279-287     (final wrap, Poseidon-BN128 hasher)
 * The boundary is the one of `prove_with_partition_witness`: the caller has built the circuit (gates, constants,
 * copy permutation) and generated the full witness; the library returns the proof in the byte format of
 * `ProofWithPublicInputs::to_bytes()` (what prove_block.rs:320-458 writes to proof.bin).
 * Gate constraints are evaluated by device code for the gate types in `zklc_plonky2_gate_type`
 * (the .go files of gnark-plonky2-verifier/plonk/gates and the .rs files of crypto/plonky2_u32/src/gates). */
enum zklc_plonky2_gate_type {
    ZKLC_GATE_NOOP = 0, ZKLC_GATE_CONSTANT /* p0 = num_consts */, ZKLC_GATE_PUBLIC_INPUT, ZKLC_GATE_ARITHMETIC /* p0 = num_ops */,
    ZKLC_GATE_ARITHMETIC_EXT /* p0 = num_ops */, ZKLC_GATE_MUL_EXT /* p0 = num_ops */, ZKLC_GATE_BASE_SUM /* p0 = num_limbs, p1 = base */,
    ZKLC_GATE_POSEIDON, ZKLC_GATE_POSEIDON_MDS, ZKLC_GATE_RANDOM_ACCESS /* p0 = bits, p1 = num_copies, p2 = num_extra_constants */,
    ZKLC_GATE_REDUCING /* p0 = num_coeffs */, ZKLC_GATE_REDUCING_EXT /* p0 = num_coeffs */, ZKLC_GATE_EXPONENTIATION /* p0 = num_power_bits */,
    ZKLC_GATE_COSET_INTERPOLATION /* p0 = subgroup_bits, p1 = degree; extra = weights then subgroup points */,
    ZKLC_GATE_U32_ARITHMETIC /* p0 = num_ops */, ZKLC_GATE_U32_ADD_MANY /* p0 = num_addends, p1 = num_ops */,
    ZKLC_GATE_U32_SUBTRACTION /* p0 = num_ops */, ZKLC_GATE_U32_RANGE_CHECK /* p0 = num_input_limbs */,
    ZKLC_GATE_COMPARISON /* p0 = num_bits, p1 = num_chunks */,
    /* crypto/plonky2_u32/src/gates/{interleave_u32,uninterleave_to_u32,uninterleave_to_b32}.rs (SHA-256 circuits); p0 = num_ops */
    ZKLC_GATE_U32_INTERLEAVE, ZKLC_GATE_UNINTERLEAVE_TO_U32, ZKLC_GATE_UNINTERLEAVE_TO_B32
};
typedef struct {
    uint32_t type;            /* zklc_plonky2_gate_type */
    uint32_t p[4];            /* gate parameters (see the enum) */
    uint32_t selector_index;  /* which selector polynomial filters this gate (common_data selectors_info) */
    uint32_t group_start, group_end; /* the selector group [start, end) this gate belongs to */
    uint32_t extra_off;       /* offset in u64 words into gate_extra */
} zklc_plonky2_gate;
#define ZKLC_HASHER_POSEIDON_GL 0u     /* PoseidonGoldilocksConfig (inner proofs) */
#define ZKLC_HASHER_POSEIDON_BN128 1u  /* PoseidonBN128GoldilocksConfig (final wrap, crypto/plonky2_bn128/src/config.rs:21-28) */
typedef struct {
    uint32_t degree_bits, num_wires, num_routed_wires, num_constants /* selectors included */, num_selectors, num_challenges;
    uint32_t rate_bits, cap_height, proof_of_work_bits, num_query_rounds;
    uint32_t quotient_degree_factor, num_partial_products, num_gate_constraints, num_public_inputs;
    uint32_t hasher, num_gates, num_arities, arity_bits[8];
} zklc_plonky2_params;
typedef struct zklc_plonky2_circuit zklc_plonky2_circuit;
/* gates: the circuit's gate list in common_data order (row i of the list = selector value i).  k_is: num_routed_wires
 * coset shifts.  constants: num_constants x 2^degree_bits values (poly-major, selectors first); sigmas: num_routed_wires x
 * 2^degree_bits values of the sigma polynomials on the subgroup.  All host pointers.  Preprocesses the circuit on the GPU
 * (constants_sigmas commitment, circuit digest) and allocates the prover's working set in HBM. */
int32_t zklc_plonky2_circuit_create(zklc_ctx *ctx, const zklc_plonky2_params *params, const zklc_plonky2_gate *gates,
                                    const uint64_t *gate_extra, uint32_t gate_extra_words, const uint64_t *k_is,
                                    const uint64_t *constants, const uint64_t *sigmas, zklc_plonky2_circuit **out);
void zklc_plonky2_circuit_destroy(zklc_plonky2_circuit *c);
/* verifier_only data: constants_sigmas_cap (2^cap_height digests of 32 bytes) and circuit_digest (32 bytes);
 * Goldilocks digests = 4 u64 LE, BN128 digests = the Fr value little-endian (PoseidonBN128HashOut::to_bytes) */
int32_t zklc_plonky2_verifier_data(zklc_plonky2_circuit *c, uint8_t *cap_out, uint8_t *digest_out);
uint64_t zklc_plonky2_proof_bytes(zklc_plonky2_circuit *c);
/* wires: num_wires x 2^degree_bits witness values (poly-major), public_inputs: num_public_inputs values (host).
 * proof_out receives zklc_plonky2_proof_bytes() bytes.  Returns ZKLC_ERR_INVALID_ARG when the witness does not
 * satisfy the copy constraints (the permutation product does not close). */
int32_t zklc_plonky2_prove(zklc_ctx *ctx, zklc_plonky2_circuit *c, const uint64_t *wires, const uint64_t *public_inputs,
                           uint8_t *proof_out, uint64_t proof_cap, uint64_t *proof_len);
/* same with the witness matrix already resident in HBM; work is enqueued on `stream` and the call returns when the
 * proof bytes are on the host (the Fiat-Shamir transcript runs on the host between kernels) */
int32_t zklc_plonky2_prove_dev(zklc_ctx *ctx, void *stream, zklc_plonky2_circuit *c, const uint64_t *d_wires,
                               const uint64_t *public_inputs, uint8_t *proof_out, uint64_t proof_cap, uint64_t *proof_len);
/* challenges of the last proof, for stage-by-stage parity tests: betas, gammas, alphas (num_challenges each), zeta (2),
 * fri_alpha (2), then one extension element per FRI reduction; returns the number of u64 written */
uint32_t zklc_plonky2_last_challenges(zklc_plonky2_circuit *c, uint64_t *out, uint32_t cap);
/* wall-clock milliseconds of the stages of the last proof: wires commit, partial products + commit, quotient + commit,
 * openings, FRI (combine + commit phase), PoW, queries + serialisation, total; returns the number of doubles written */
uint32_t zklc_plonky2_last_timings(zklc_plonky2_circuit *c, double *out_ms, uint32_t cap);
/* PoseidonGate witness rows (host function, no GPU): inputs n x 12, swap n (0/1, NULL = all 0) -> rows n x 135
 * in the wire layout of gnark-plonky2-verifier/plonk/gates/poseidon_gate.go:27-82 */
int32_t zklc_poseidon_gl_gate_rows(const uint64_t *inputs, const uint64_t *swap, uint32_t n, uint64_t *rows);
/* out[i] = a[i] * b[i] in the Goldilocks field (host function; canonical inputs): the circuit builder's sigma values */
void zklc_gl_mul_vec(const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n);
/* copy classes of a circuit under construction (host function): union-find over n_pairs `connect(a, b)` records given as dense
 * target indices; root_out[i] = the smallest index of i's class.  The native part of the host circuit builder (plonky2's
 * `wire_partition`, reached from every gadget of crypto/plonky2_ed25519/src/gadgets). */
int32_t zklc_host_copy_classes(const int64_t *ia, const int64_t *ib, uint64_t n_pairs, uint64_t n_keys, int64_t *root_out);
/* Poseidon-Goldilocks parameters (gnark-plonky2-verifier/poseidon/goldilocks_constants.go): all round constants (30 x 12),
 * the optimised partial-round constants (first layer 12, one per round 22), MDS circulant + diagonal.  Used by the host
 * circuit builder to restate PoseidonGate's constraints in-circuit (recursive verifier). */
void zklc_poseidon_gl_constants(uint64_t *rc360, uint64_t *fp_first12, uint64_t *fp_rc22, uint64_t *mds_circ12, uint64_t *mds_diag12);
/* Native witness generation (host, multi-threaded; no GPU): executes the generator program of a circuit built by the host
 * builder -- the witness generators of crypto/plonky2_ed25519/src/gadgets/nonnative.rs:447-705, gadgets/curve.rs:327-370,
 * crypto/plonky2_ecdsa/src/gadgets/biguint.rs:417-470 and the gate generators of crypto/plonky2_u32/src/gates -- for
 * n_witnesses partial witnesses.  code (u32 words) = [opcode, n_params, n_in, n_out, input slots.., output slots..]*, the
 * parameters of all instructions consecutive in `params`; a slot = one copy class of the circuit
 * (zklc_amd/plonky2/builder.py `witness_program`).  wire_slot / wire_index (= col * n_rows + row): the wire cells.
 * wires_out: n_witnesses x num_wires x n_rows u64, only the circuit's wire cells are written (zero-fill once, reuse);
 * status[i] != 0 when witness i does not exist (e.g. an invalid signature: "copy constraint violated"), message in
 * err_out[200 i ..]. */
int32_t zklc_plonky2_witness_run(const uint32_t *code, uint64_t code_len, const int64_t *params, uint32_t n_slots,
                                 const uint32_t *input_slots, uint32_t n_inputs, const uint64_t *input_values, uint32_t n_witnesses,
                                 const uint32_t *wire_slot, const uint32_t *wire_index, uint64_t n_wire_entries, uint32_t num_wires,
                                 uint32_t n_rows, uint64_t *wires_out, const uint32_t *pi_slots, uint32_t n_pi, uint64_t *pi_out,
                                 int32_t *status, char *err_out, uint32_t threads);
/* the interpreter keeps its per-thread state (value arrays, one per slot) cached across calls; this frees the cache */
void zklc_plonky2_witness_release(void);

/* Witness generation ON THE GPU: the same program, levelled by data dependence and executed by the device for a
 * batch of up to 64 witnesses -- one lane per (instruction, witness), slot values stored as val[slot][witness]
 * (csrc/plonky2_witness_dev.hip).  The wire matrices are written in HBM in the prover's layout (what zklc_plonky2_prove_dev
 * takes), so the 490 MB per Ed25519 signature never cross PCIe and no host core is on the per-signature path.
 * Replaces the generators that run inside `CircuitData::prove` at near_bft_finality/src/prove_crypto/ed25519.rs:60,100 and
 * recursion.rs:95 (crypto/plonky2_ed25519/src/gadgets/nonnative.rs:447-705, gadgets/curve.rs:327-370, ...).
 * program_create copies and schedules the program once per circuit (arguments as for zklc_plonky2_witness_run);
 * run_dev: input_values host n_witnesses x n_inputs; d_wires DEVICE n_witnesses x num_wires x n_rows (only the circuit's wire
 * cells are written: zero-fill the buffer once); pi_out / status / err_out (optional, 200 bytes per witness) are host arrays,
 * valid on return (the call synchronises `stream`).  status[i] != 0: witness i does not exist (invalid signature, ...). */
typedef struct zklc_witness_program zklc_witness_program;
int32_t zklc_plonky2_witness_program_create(zklc_ctx *ctx, const uint32_t *code, uint64_t code_len, const int64_t *params,
                                            uint64_t n_params, uint32_t n_slots, const uint32_t *input_slots, uint32_t n_inputs,
                                            const uint32_t *wire_slot, const uint32_t *wire_index, uint64_t n_wire_entries,
                                            uint32_t num_wires, uint32_t n_rows, const uint32_t *pi_slots, uint32_t n_pi,
                                            zklc_witness_program **out);
void zklc_plonky2_witness_program_destroy(zklc_witness_program *p);
/* instructions, dependence levels, and the kernel launches a batch of n_witnesses takes (runs of small levels share a launch) */
int32_t zklc_plonky2_witness_program_info(zklc_witness_program *p, uint32_t n_witnesses, uint64_t *n_instr, uint32_t *n_levels,
                                          uint32_t *n_launches);
int32_t zklc_plonky2_witness_run_dev(zklc_ctx *ctx, void *stream, zklc_witness_program *p, const uint64_t *input_values,
                                     uint32_t n_witnesses, uint64_t *d_wires, uint64_t *pi_out, int32_t *status, char *err_out);

/* ---- (b'') circuit container: the hand-off between `builder.build()` and `data.prove(pw)` ------------------------
 * The reference builds a circuit once and proves with it many times: `get_ed25519_circuit_targets` /
 * `ed25519_proof_reuse_circuit` (near_bft_finality/src/prove_crypto/ed25519.rs:18-42 build, :44-66 prove; the call at :60),
 * `recursive_proof` (recursion.rs:36-94 build, :95 prove).  The container is that seam as a FILE: a flat little-endian list of
 * tagged sections holding exactly the arguments of zklc_plonky2_circuit_create and zklc_plonky2_witness_program_create, written
 * by whoever ran the circuit builder (a Rust shim beside plonky2's CircuitBuilder: INTEGRATION.md; this repo's Python mirror:
 * zklc_amd/plonky2/container.py) and loaded by any caller of this ABI (tests/c_abi/prove_from_file.c proves from two files and
 * nothing else).  Layout (csrc/container.cpp): 64-byte header (magic "ZKLCCIRC", version, section count, file size, table
 * checksum), one 32-byte table entry per section (tag, element size, offset, bytes, checksum), payload sections on 64-byte
 * boundaries.  The file functions are host code: they work without a GPU. */
#define ZKLC_CONTAINER_VERSION 1u
#define ZKLC_CONTAINER_VERIFY 1u      /* zklc_container_open flag: also check every section's checksum (reads the whole file) */
enum zklc_container_tag {
    ZKLC_SEC_PARAMS = 1,           /* one zklc_plonky2_params (hasher = the builder's default; create_from_container can override it) */
    ZKLC_SEC_GATES = 2,            /* zklc_plonky2_gate[num_gates] */
    ZKLC_SEC_GATE_EXTRA = 3,       /* u64[]: gate_extra (absent when no gate has a table) */
    ZKLC_SEC_K_IS = 4,             /* u64[num_routed_wires] */
    ZKLC_SEC_CONSTANTS = 5,        /* u64[num_constants x 2^degree_bits], selectors first */
    ZKLC_SEC_SIGMAS = 6,           /* u64[num_routed_wires x 2^degree_bits] */
    ZKLC_SEC_WP_DIMS = 16,         /* one zklc_witness_dims: the scalar arguments of zklc_plonky2_witness_program_create */
    ZKLC_SEC_WP_CODE = 17,         /* u32[code_len] */
    ZKLC_SEC_WP_PARAMS = 18,       /* i64[n_params] */
    ZKLC_SEC_WP_INPUT_SLOTS = 19,  /* u32[n_inputs] */
    ZKLC_SEC_WP_WIRE_SLOT = 20,    /* u32[n_wire_entries] */
    ZKLC_SEC_WP_WIRE_INDEX = 21,   /* u32[n_wire_entries] */
    ZKLC_SEC_WP_PI_SLOTS = 22,     /* u32[n_pi] */
    ZKLC_SEC_INPUT_VALUES = 32,    /* u64[n_witnesses x n_inputs]: a witness-input file (the PartialWitness of ed25519.rs:54-59 /
                                      recursion.rs:44-92 in the order of the program's inputs) */
    ZKLC_SEC_HOST_FIRST = 0x1000   /* tags from here on belong to the writer (target names, JSON notes); the library ignores them */
};
typedef struct {
    uint64_t code_len, n_params, n_wire_entries;
    uint32_t n_slots, n_inputs, num_wires, n_rows, n_pi, reserved;
} zklc_witness_dims;
typedef struct {
    uint32_t tag;         /* zklc_container_tag or a writer's own tag >= ZKLC_SEC_HOST_FIRST; one section per tag */
    uint32_t elem_bytes;  /* size of one element (the section is an array of them): 1, 4, 8 or a struct size */
    const void *data;
    uint64_t bytes;
} zklc_container_entry;
typedef struct zklc_container zklc_container;
/* writes `path` atomically (temporary file beside it + rename).  ZKLC_ERR_IO when the directory cannot be written. */
int32_t zklc_container_write(const char *path, const zklc_container_entry *entries, uint32_t n_entries);
/* the circuit of zklc_plonky2_circuit_create and -- when `dims` is not NULL -- the witness program of
 * zklc_plonky2_witness_program_create as one container; extra_entries: the writer's own sections (tags >= ZKLC_SEC_HOST_FIRST) */
int32_t zklc_plonky2_container_write(const char *path, const zklc_plonky2_params *params, const zklc_plonky2_gate *gates,
                                     const uint64_t *gate_extra, uint32_t gate_extra_words, const uint64_t *k_is,
                                     const uint64_t *constants, const uint64_t *sigmas, const zklc_witness_dims *dims,
                                     const uint32_t *code, const int64_t *wparams, const uint32_t *input_slots,
                                     const uint32_t *wire_slot, const uint32_t *wire_index, const uint32_t *pi_slots,
                                     const zklc_container_entry *extra_entries, uint32_t n_extra_entries);
/* maps the file read-only and checks header, size and section table (with ZKLC_CONTAINER_VERIFY: every section's checksum).
 * The data pointers of the entries stay valid until zklc_container_close. */
int32_t zklc_container_open(const char *path, uint32_t flags, zklc_container **out);
void zklc_container_close(zklc_container *c);
uint32_t zklc_container_count(const zklc_container *c);
int32_t zklc_container_entry_at(const zklc_container *c, uint32_t i, zklc_container_entry *out);
int32_t zklc_container_find(const zklc_container *c, uint32_t tag, zklc_container_entry *out);
/* after the circuit is on the GPU the mapped pages are dead weight: hand them back (the mapping stays valid) */
void zklc_container_release_pages(const zklc_container *c);
/* the parameter blocks of a circuit container after the consistency checks the create functions below run (either may be NULL) */
int32_t zklc_plonky2_container_params(const zklc_container *c, zklc_plonky2_params *params_out, zklc_witness_dims *dims_out);
/* zklc_plonky2_circuit_create / zklc_plonky2_witness_program_create with the arguments taken from the container's sections,
 * after checking that every section has the size the parameters imply and every index is in range (ZKLC_ERR_FORMAT otherwise).
 * hasher: ZKLC_HASHER_* or -1 = the one stored in the file.  The container may be closed as soon as the call returns. */
int32_t zklc_plonky2_circuit_create_from_container(zklc_ctx *ctx, const zklc_container *c, int32_t hasher, zklc_plonky2_circuit **out);
int32_t zklc_plonky2_witness_program_create_from_container(zklc_ctx *ctx, const zklc_container *c, zklc_witness_program **out);
/* zklc_plonky2_witness_run (host interpreter, no GPU) over the container's program */
int32_t zklc_plonky2_witness_run_from_container(const zklc_container *c, const uint64_t *input_values, uint32_t n_witnesses,
                                                uint64_t *wires_out, uint64_t *pi_out, int32_t *status, char *err_out, uint32_t threads);

/* ---- (c) BN254 ---------------------------------------------------------------
 * G1 multi-scalar multiplication sum_i scalars[i] * points[i].
 * Replaces gnark-crypto `bn254.G1Affine.MultiExp` (un-vendored; gnark-plonky2-verifier/go.mod:9)
 * called from `groth16.Prove` at gnark-plonky2-verifier/cmd/web-api.go:77.
 * points: n affine points in gnark-crypto's memory layout = x then y, each 4 little-endian
 * u64 limbs in Montgomery form (x * 2^256 mod p); (0, 0) is the point at infinity.
 * scalars: n x 4 little-endian u64, REGULAR (non-Montgomery) form; expected reduced (< r) -- a scalar >= r is reduced by the
 * kernel (the points have order r), it is not an error.
 * out_affine: 8 u64 in the same layout, canonical; *out_is_infinity = 1 when the sum is
 * the point at infinity (then out_affine is zero).  Pointers must be 16-byte aligned. */
int32_t zklc_bn254_g1_msm(zklc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, uint64_t n, uint64_t *out_affine,
                          uint32_t *out_is_infinity);
uint64_t zklc_bn254_g1_msm_workspace_bytes(uint64_t n);
int32_t zklc_bn254_g1_msm_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, const uint64_t *d_scalars, uint64_t n,
                              uint64_t *d_out_affine, uint32_t *d_out_is_infinity, void *d_workspace, uint64_t workspace_bytes);

/* G2 multi-scalar multiplication (the B2 term of a Groth16 proof): replaces gnark-crypto `bn254.G2Affine.MultiExp`.
 * points: n affine G2 points = X.A0, X.A1, Y.A0, Y.A1 (4 x 4 little-endian u64, Montgomery form; all zero = infinity);
 * scalars as for the G1 MSM; out_affine: 16 u64 in the same layout. */
int32_t zklc_bn254_g2_msm(zklc_ctx *ctx, const uint64_t *points, const uint64_t *scalars, uint64_t n, uint64_t *out_affine,
                          uint32_t *out_is_infinity);
uint64_t zklc_bn254_g2_msm_workspace_bytes(uint64_t n);
int32_t zklc_bn254_g2_msm_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, const uint64_t *d_scalars, uint64_t n,
                              uint64_t *d_out_affine, uint32_t *d_out_is_infinity, void *d_workspace, uint64_t workspace_bytes);

/* Fixed-base form of the two multi-exponentiations (round 5).  `groth16.Prove` (gnark-plonky2-verifier/cmd/web-api.go:77) multiplies
 * the SAME bases -- the proving key's pk.G1.A / pk.G1.B / pk.G1.K / pk.G1.Z and pk.G2.B -- by every proof's scalars: a table of
 * 2^(c w) * P_i for every window w (windows x n packed affine records: 64 bytes per G1 record, 128 per G2; c and the number of
 * windows are the library's choice for n and are recorded in the table's 256-byte header) is built ONCE per key, and every later
 * multi-exponentiation adds table points into one bucket set shared by all windows: no closing doublings, one bucket reduction.
 * Result: the same canonical affine point as zklc_bn254_g{1,2}_msm[_dev] on (points, scalars), bit for bit.
 *   *_table_bytes(n)   bytes of device memory the table of n bases needs;
 *   *_table_dev        builds it from n affine points in HBM (layout as for the plain form) into d_table (256-byte aligned);
 *   *_fixed_dev        the multi-exponentiation of the table's bases by d_scalars (n x 4 u64, regular form); n must be the
 *                      table's n; workspace as zklc_bn254_g{1,2}_msm_workspace_bytes(n).  ZKLC_ERR_INVALID_ARG if d_table is
 *                      not a table of this group for n bases.  Enqueue only, except a 20-byte header read. */
uint64_t zklc_bn254_g1_msm_fixed_table_bytes(uint64_t n);
int32_t zklc_bn254_g1_msm_fixed_table_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, uint64_t n, void *d_table,
                                          uint64_t table_bytes);
int32_t zklc_bn254_g1_msm_fixed_dev(zklc_ctx *ctx, void *stream, const void *d_table, const uint64_t *d_scalars, uint64_t n,
                                    uint64_t *d_out_affine, uint32_t *d_out_is_infinity, void *d_workspace, uint64_t workspace_bytes);
uint64_t zklc_bn254_g2_msm_fixed_table_bytes(uint64_t n);
int32_t zklc_bn254_g2_msm_fixed_table_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_points, uint64_t n, void *d_table,
                                          uint64_t table_bytes);
int32_t zklc_bn254_g2_msm_fixed_dev(zklc_ctx *ctx, void *stream, const void *d_table, const uint64_t *d_scalars, uint64_t n,
                                    uint64_t *d_out_affine, uint32_t *d_out_is_infinity, void *d_workspace, uint64_t workspace_bytes);

/* Pairing-product checks: is_one[b] = (prod_{i<k} e(P_{b,i}, Q_{b,i}) == 1) for `batch` independent checks.
 * Replaces gnark-crypto `bn254.PairingCheck` behind `groth16.Verify` (gnark-plonky2-verifier/cmd/web-api.go:84) and the EVM
 * pairing precompile the reference's Solidity verifier calls (contracts/hardhat/contracts/Verifier.sol:503-548: k = 4,
 * e(A,B) e(C,-delta) e(alpha,-beta) e(L_pub,-gamma)).  g1: batch x k affine G1 points (8 u64 each), g2: batch x k affine G2
 * points (16 u64 each), gnark-crypto memory layout (Montgomery); an all-zero point is the point at infinity.
 * gt_out (optional, may be NULL): the reduced pairing product of every check, 12 Fp coefficients in gnark-crypto's E12 order
 * (C0.B0.A0, C0.B0.A1, C0.B1.A0, ... C1.B2.A1), 48 u64 per check, exact exponent (p^12 - 1)/r.  Points must be on their
 * curves and in the r-torsion subgroups (as the precompile requires); this is NOT checked -- the same contract as gnark-crypto's
 * `bn254.PairingCheck`, whose caller `groth16.Verify` validates the proof's points itself (`proof.isValid()`, unchanged Go code
 * above a cgo shim); INTEGRATION.md states the precondition at the call site. */
int32_t zklc_bn254_pairing_check(zklc_ctx *ctx, const uint64_t *g1, const uint64_t *g2, uint32_t k, uint32_t batch,
                                 uint32_t *is_one, uint64_t *gt_out);
int32_t zklc_bn254_pairing_check_dev(zklc_ctx *ctx, void *stream, const uint64_t *d_g1, const uint64_t *d_g2, uint32_t k,
                                     uint32_t batch, uint32_t *d_is_one, uint64_t *d_gt_out);

/* NTT over the BN254 scalar field Fr.  Replaces gnark-crypto `fft.Domain.FFT / FFTInverse` (ecc/bn254/fr/fft, un-vendored)
 * inside `groth16.Prove` (gnark-plonky2-verifier/cmd/web-api.go:77).  data: 2^log_n elements in gnark-crypto's memory layout
 * (x * 2^256 mod r, 4 little-endian u64), transformed in place.  values[k] = sum_j coeffs[j] w^(jk), w = rootOfUnity^(2^28/n);
 * flags as for zklc_gl_ntt (ZKLC_NTT_INVERSE includes 1/n).  coset = 1: evaluate on / interpolate from the coset 5 * <w>
 * (gnark `fft.OnCoset()`, FrMultiplicativeGen = 5). */
int32_t zklc_bn254_fr_ntt(zklc_ctx *ctx, uint64_t *data, uint32_t log_n, uint32_t flags, uint32_t coset);
uint64_t zklc_bn254_fr_ntt_workspace_bytes(uint32_t log_n);
int32_t zklc_bn254_fr_ntt_dev(zklc_ctx *ctx, void *stream, uint64_t *d_data, uint32_t log_n, uint32_t flags, uint32_t coset,
                              void *d_workspace, uint64_t workspace_bytes);

/* d_a[i] = (d_a[i] * d_b[i] - d_c[i]) * scale over Fr (device arrays of n elements, gnark layout; scale: 4 u64, host): the
 * pointwise step of the quotient polynomial in `groth16.Prove` (gnark backend/groth16/bn254/prove.go `computeH`, un-vendored),
 * between the three coset FFTs and the coset inverse FFT; scale = 1 / (5^n - 1). */
int32_t zklc_bn254_fr_mul_sub_scale_dev(zklc_ctx *ctx, void *stream, uint64_t *d_a, const uint64_t *d_b, const uint64_t *d_c,
                                        const uint64_t *scale, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* ZKLC_H */
