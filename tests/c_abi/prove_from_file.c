/* prove_from_file -- a native caller of the C ABI and nothing else.
 *
 * Plays the part of the reference between `builder.build()` and `data.prove(pw)`
 * (near_bft_finality/src/prove_crypto/ed25519.rs:26-39 -> :54-60, recursion.rs:94 -> :95): it is handed a circuit container
 * (include/zklc.h section b'': what a Rust shim beside plonky2's CircuitBuilder would emit) and a witness-input file (the
 * PartialWitness as u64 values in the order of the program's inputs), creates the circuit and the witness program on the GPU,
 * generates the witness there, proves, and writes `ProofWithPublicInputs::to_bytes()` -- the proof.bin of
 * bin/prove_block.rs:320-458.  No Python, no HIP headers: plain C against libzklc_mi355.so.
 *
 *     prove_from_file <circuit.zkcc> <inputs.zkcc> <proof_out.bin> [--hasher 0|1] [--host-witness] [--repeat K]
 *
 * Exit status 0 = proofs written (with --repeat K the K proofs must be byte-identical: the prover is deterministic);
 * 2 = usage, 3 = a zklc call failed (its name and status go to stderr).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "zklc.h"

#define CHECK(call)                                                                                        \
    do {                                                                                                   \
        int32_t rc__ = (call);                                                                             \
        if (rc__ != ZKLC_OK) {                                                                             \
            fprintf(stderr, "prove_from_file: %s -> %d (%s)%s%s\n", #call, rc__, zklc_strerror(rc__),      \
                    ctx ? " / " : "", ctx ? zklc_last_hip_error(ctx) : "");                                \
            return 3;                                                                                      \
        }                                                                                                  \
    } while (0)

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec / 1e6;
}

int main(int argc, char **argv) {
    zklc_ctx *ctx = NULL;
    int hasher = -1, host_witness = 0, repeat = 1;
    if (argc < 4) {
        fprintf(stderr, "usage: %s <circuit container> <witness-input file> <proof out> [--hasher 0|1] [--host-witness] [--repeat K]\n", argv[0]);
        return 2;
    }
    for (int i = 4; i < argc; i++) {
        if (!strcmp(argv[i], "--hasher") && i + 1 < argc) hasher = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--host-witness")) host_witness = 1;
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
        else {
            fprintf(stderr, "unknown option %s\n", argv[i]);
            return 2;
        }
    }
    if (repeat < 1) repeat = 1;

    /* ---- the two files (host only: no GPU touched yet) */
    zklc_container *circ = NULL, *inp = NULL;
    CHECK(zklc_container_open(argv[1], ZKLC_CONTAINER_VERIFY, &circ));
    CHECK(zklc_container_open(argv[2], ZKLC_CONTAINER_VERIFY, &inp));
    zklc_plonky2_params P;
    zklc_witness_dims D;
    CHECK(zklc_plonky2_container_params(circ, &P, &D));
    zklc_container_entry in;
    CHECK(zklc_container_find(inp, ZKLC_SEC_INPUT_VALUES, &in));
    if (in.elem_bytes != 8 || D.n_inputs == 0 || in.bytes % (8ull * D.n_inputs)) {
        fprintf(stderr, "prove_from_file: the input file holds %llu bytes, the program has %u inputs\n", (unsigned long long)in.bytes, D.n_inputs);
        return 3;
    }
    const uint32_t n_wit = (uint32_t)(in.bytes / (8ull * D.n_inputs));
    const uint64_t n_rows = 1ull << P.degree_bits, cells = (uint64_t)P.num_wires * n_rows;
    if (n_wit == 0 || n_wit > 64 || D.num_wires != P.num_wires || D.n_rows != n_rows || D.n_pi != P.num_public_inputs) {
        fprintf(stderr, "prove_from_file: %u witnesses / program and circuit disagree\n", n_wit);
        return 3;
    }
    fprintf(stderr, "circuit: 2^%u rows x %u wires, %u gate types, %u public inputs; program: %llu code words, %u slots, %u inputs; %u witness(es)\n",
            P.degree_bits, P.num_wires, P.num_gates, P.num_public_inputs, (unsigned long long)D.code_len, D.n_slots, D.n_inputs, n_wit);

    /* ---- GPU: context, circuit, witness program */
    double t0 = now_ms();
    CHECK(zklc_init(&ctx, 0));
    zklc_plonky2_circuit *c = NULL;
    CHECK(zklc_plonky2_circuit_create_from_container(ctx, circ, hasher, &c));
    const uint64_t plen = zklc_plonky2_proof_bytes(c);
    uint8_t *proof = (uint8_t *)malloc(plen * (size_t)n_wit), *again = (uint8_t *)malloc(plen);
    uint64_t *pis = (uint64_t *)calloc((size_t)n_wit * (P.num_public_inputs ? P.num_public_inputs : 1), 8);
    int32_t *status = (int32_t *)calloc(n_wit, 4);
    char *err = (char *)calloc(n_wit, 200);
    if (!proof || !again || !pis || !status || !err) return 3;
    double t1 = now_ms();

    void *d_wires = NULL;
    uint64_t *h_wires = NULL;
    if (host_witness) {
        /* the host interpreter (no GPU) + the host-pointer prover entry: the path a caller without device buffers takes */
        h_wires = (uint64_t *)calloc((size_t)n_wit * cells, 8);
        if (!h_wires) return 3;
        CHECK(zklc_plonky2_witness_run_from_container(circ, (const uint64_t *)in.data, n_wit, h_wires, pis, status, err, 4));
    } else {
        zklc_witness_program *wp = NULL;
        CHECK(zklc_plonky2_witness_program_create_from_container(ctx, circ, &wp));
        CHECK(zklc_device_alloc(ctx, (uint64_t)n_wit * cells * 8, &d_wires));            /* zero-filled */
        CHECK(zklc_plonky2_witness_run_dev(ctx, zklc_stream(ctx), wp, (const uint64_t *)in.data, n_wit, (uint64_t *)d_wires, pis, status, err));
        zklc_plonky2_witness_program_destroy(wp);
    }
    for (uint32_t w = 0; w < n_wit; w++)
        if (status[w]) {
            fprintf(stderr, "prove_from_file: witness %u does not exist: %s\n", w, err + 200 * w);
            return 3;
        }
    zklc_container_close(inp);
    zklc_container_close(circ);            /* everything the GPU needs has been uploaded */
    double t2 = now_ms();

    /* ---- prove */
    for (uint32_t w = 0; w < n_wit; w++) {
        uint64_t got = 0;
        const uint64_t *pi_w = pis + (size_t)w * P.num_public_inputs;
        if (host_witness) CHECK(zklc_plonky2_prove(ctx, c, h_wires + (size_t)w * cells, pi_w, proof + plen * w, plen, &got));
        else CHECK(zklc_plonky2_prove_dev(ctx, zklc_stream(ctx), c, (const uint64_t *)d_wires + (size_t)w * cells, pi_w, proof + plen * w, plen, &got));
        if (got != plen) {
            fprintf(stderr, "prove_from_file: proof of %llu bytes, expected %llu\n", (unsigned long long)got, (unsigned long long)plen);
            return 3;
        }
    }
    double t3 = now_ms();
    for (int r = 1; r < repeat; r++) {
        uint64_t got = 0;
        if (host_witness) CHECK(zklc_plonky2_prove(ctx, c, h_wires, pis, again, plen, &got));
        else CHECK(zklc_plonky2_prove_dev(ctx, zklc_stream(ctx), c, (const uint64_t *)d_wires, pis, again, plen, &got));
        if (got != plen || memcmp(again, proof, plen)) {
            fprintf(stderr, "prove_from_file: repeat %d gave different bytes\n", r);
            return 3;
        }
    }
    double t4 = now_ms();
    double st[8];
    uint32_t ns = zklc_plonky2_last_timings(c, st, 8);

    FILE *f = fopen(argv[3], "wb");
    if (!f || fwrite(proof, 1, plen * (size_t)n_wit, f) != plen * (size_t)n_wit || fclose(f)) {
        fprintf(stderr, "prove_from_file: cannot write %s\n", argv[3]);
        return 3;
    }
    uint8_t cap[32 * 64], dig[32];
    if (P.cap_height <= 6) {
        CHECK(zklc_plonky2_verifier_data(c, cap, dig));
        fprintf(stderr, "circuit_digest:");
        for (int i = 0; i < 32; i++) fprintf(stderr, "%02x", dig[i]);
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "create %.1f ms, witness (%s) %.1f ms, %u proof(s) %.1f ms", t1 - t0, host_witness ? "host" : "device", t2 - t1, n_wit, t3 - t2);
    if (repeat > 1) fprintf(stderr, ", then %.2f ms per proof over %d repeats (byte-identical)", (t4 - t3) / (repeat - 1), repeat - 1);
    if (ns >= 8) fprintf(stderr, "; last proof: wires %.2f / Z %.2f / quotient %.2f / openings %.2f / FRI %.2f ms", st[0], st[1], st[2], st[3], st[4]);
    fprintf(stderr, "\n%llu proof bytes x %u -> %s\n", (unsigned long long)plen, n_wit, argv[3]);

    if (d_wires) CHECK(zklc_device_free(ctx, d_wires));
    free(h_wires);
    zklc_plonky2_circuit_destroy(c);
    zklc_destroy(ctx);
    free(proof), free(again), free(pis), free(status), free(err);
    return 0;
}
