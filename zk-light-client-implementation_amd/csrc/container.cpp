// Circuit container (include/zklc.h, section b''): the language-neutral hand-off between `builder.build()` and
// `data.prove(pw)` -- near_bft_finality/src/prove_crypto/ed25519.rs:26-39 (build once per message length) / :60 (prove),
// recursion.rs:94 / :95.  A flat little-endian file of tagged sections that hold EXACTLY the arguments of
// zklc_plonky2_circuit_create and zklc_plonky2_witness_program_create, so that whoever built the circuit (a Rust shim next to
// plonky2's CircuitBuilder, or this repo's Python mirror) and whoever proves with it (any caller of this C ABI, e.g.
// tests/c_abi/prove_from_file.c) need to share nothing but this file.
//
//   header   64 bytes : magic "ZKLCCIRC", u32 version, u32 n_sections, u64 file_bytes, u64 table_hash, 32 bytes reserved (0)
//   table    n x 32   : u32 tag, u32 elem_bytes, u64 offset, u64 bytes, u64 hash
//   payload           : every section starts on a 64-byte boundary; gaps are zero
//
// hash = four interleaved FNV-1a-64 lanes over the little-endian u64 words of the section (the tail zero-padded to a word),
// folded as ((h0 * P + h1) * P + h2) * P + h3 and mixed with the byte length; table_hash = the same over the table.
// Host code only (no GPU): the file functions work on a machine without a device; the *_from_container entry points forward
// to the ordinary create functions after checking that every section has the size the parameters imply.
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <new>
#include <string>
#include <vector>
#include "../../include/zklc.h"

namespace {
typedef uint64_t u64;
typedef uint32_t u32;

const char MAGIC[8] = {'Z', 'K', 'L', 'C', 'C', 'I', 'R', 'C'};
const u64 FNV_P = 0x100000001b3ULL, FNV_O = 0xcbf29ce484222325ULL;

struct file_header {
    char magic[8];
    u32 version, n_sections;
    u64 file_bytes, table_hash;
    u64 reserved[4];
};
struct file_section {
    u32 tag, elem_bytes;
    u64 offset, bytes, hash;
};
static_assert(sizeof(file_header) == 64 && sizeof(file_section) == 32, "container layout");

u64 hash_bytes(const void *data, u64 bytes) {
    const unsigned char *p = (const unsigned char *)data;
    u64 h[4] = {FNV_O, FNV_O ^ 1, FNV_O ^ 2, FNV_O ^ 3};
    u64 words = bytes / 8, i = 0;
    for (; i + 4 <= words; i += 4) {
        u64 w[4];
        memcpy(w, p + 8 * i, 32);
        h[0] = (h[0] ^ w[0]) * FNV_P;
        h[1] = (h[1] ^ w[1]) * FNV_P;
        h[2] = (h[2] ^ w[2]) * FNV_P;
        h[3] = (h[3] ^ w[3]) * FNV_P;
    }
    for (; i < words; i++) {
        u64 w;
        memcpy(&w, p + 8 * i, 8);
        h[i & 3] = (h[i & 3] ^ w) * FNV_P;
    }
    if (bytes & 7) {
        u64 w = 0;
        memcpy(&w, p + 8 * words, bytes & 7);
        h[words & 3] = (h[words & 3] ^ w) * FNV_P;
    }
    u64 r = ((h[0] * FNV_P + h[1]) * FNV_P + h[2]) * FNV_P + h[3];
    return (r ^ bytes) * FNV_P;
}

u64 align64(u64 x) { return (x + 63) & ~(u64)63; }

bool write_all(int fd, const void *p, u64 n) {
    const char *q = (const char *)p;
    while (n) {
        ssize_t k = write(fd, q, n > (1u << 30) ? (1u << 30) : n);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        q += k;
        n -= (u64)k;
    }
    return true;
}
}  // namespace

struct zklc_container {
    void *map = nullptr;
    u64 bytes = 0;
    std::vector<zklc_container_entry> entries;
};

extern "C" int32_t zklc_container_write(const char *path, const zklc_container_entry *entries, uint32_t n_entries) {
    if (!path || (!entries && n_entries) || n_entries > 4096) return ZKLC_ERR_INVALID_ARG;
    std::vector<file_section> tab(n_entries);
    u64 off = align64(sizeof(file_header) + (u64)n_entries * sizeof(file_section));
    for (u32 i = 0; i < n_entries; i++) {
        const zklc_container_entry &e = entries[i];
        if ((e.bytes && !e.data) || e.elem_bytes == 0 || e.bytes % e.elem_bytes) return ZKLC_ERR_INVALID_ARG;
        for (u32 j = 0; j < i; j++)
            if (entries[j].tag == e.tag) return ZKLC_ERR_INVALID_ARG;          // one section per tag
        tab[i] = {e.tag, e.elem_bytes, off, e.bytes, hash_bytes(e.data, e.bytes)};
        off = align64(off + e.bytes);
    }
    file_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, MAGIC, 8);
    h.version = ZKLC_CONTAINER_VERSION;
    h.n_sections = n_entries;
    h.file_bytes = off;
    h.table_hash = hash_bytes(tab.data(), tab.size() * sizeof(file_section));
    // written beside the target and renamed over it: a reader never sees a partial file, concurrent writers never interleave
    std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
    int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return ZKLC_ERR_IO;
    static const char zeros[64] = {};
    bool ok = write_all(fd, &h, sizeof h) && write_all(fd, tab.data(), tab.size() * sizeof(file_section));
    u64 pos = sizeof h + tab.size() * sizeof(file_section);
    for (u32 i = 0; ok && i < n_entries; i++) {
        ok = write_all(fd, zeros, tab[i].offset - pos) && write_all(fd, entries[i].data, entries[i].bytes);
        pos = tab[i].offset + entries[i].bytes;
    }
    ok = ok && write_all(fd, zeros, off - pos);
    ok = (close(fd) == 0) && ok;
    if (ok && rename(tmp.c_str(), path) != 0) ok = false;
    if (!ok) {
        unlink(tmp.c_str());
        return ZKLC_ERR_IO;
    }
    return ZKLC_OK;
}

extern "C" void zklc_container_close(zklc_container *c) {
    if (!c) return;
    if (c->map) munmap(c->map, c->bytes);
    delete c;
}

extern "C" int32_t zklc_container_open(const char *path, uint32_t flags, zklc_container **out) {
    if (!path || !out) return ZKLC_ERR_INVALID_ARG;
    *out = nullptr;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return ZKLC_ERR_IO;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < (off_t)sizeof(file_header)) {
        close(fd);
        return ZKLC_ERR_FORMAT;
    }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ZKLC_ERR_IO;
    zklc_container *c = new (std::nothrow) zklc_container();
    if (!c) {
        munmap(m, (size_t)st.st_size);
        return ZKLC_ERR_OOM;
    }
    c->map = m;
    c->bytes = (u64)st.st_size;
    const file_header *h = (const file_header *)m;
    auto fail = [&](int32_t rc) {
        zklc_container_close(c);
        return rc;
    };
    if (memcmp(h->magic, MAGIC, 8) != 0) return fail(ZKLC_ERR_FORMAT);
    if (h->version != ZKLC_CONTAINER_VERSION) return fail(ZKLC_ERR_FORMAT);
    if (h->file_bytes != c->bytes || h->n_sections > 4096) return fail(ZKLC_ERR_FORMAT);           // truncated or extended
    u64 tab_bytes = (u64)h->n_sections * sizeof(file_section);
    if (sizeof(file_header) + tab_bytes > c->bytes) return fail(ZKLC_ERR_FORMAT);
    const file_section *tab = (const file_section *)((const char *)m + sizeof(file_header));
    if (hash_bytes(tab, tab_bytes) != h->table_hash) return fail(ZKLC_ERR_FORMAT);
    for (u32 i = 0; i < h->n_sections; i++) {
        const file_section &s = tab[i];
        if (s.offset % 64 || s.offset < sizeof(file_header) + tab_bytes || s.offset > c->bytes || s.bytes > c->bytes - s.offset ||
            s.elem_bytes == 0 || s.bytes % s.elem_bytes)
            return fail(ZKLC_ERR_FORMAT);
        for (u32 j = 0; j < i; j++)
            if (tab[j].tag == s.tag) return fail(ZKLC_ERR_FORMAT);
        if ((flags & ZKLC_CONTAINER_VERIFY) && hash_bytes((const char *)m + s.offset, s.bytes) != s.hash) return fail(ZKLC_ERR_FORMAT);
        c->entries.push_back({s.tag, s.elem_bytes, (const char *)m + s.offset, s.bytes});
    }
    *out = c;
    return ZKLC_OK;
}

extern "C" uint32_t zklc_container_count(const zklc_container *c) { return c ? (uint32_t)c->entries.size() : 0; }

extern "C" int32_t zklc_container_entry_at(const zklc_container *c, uint32_t i, zklc_container_entry *out) {
    if (!c || !out || i >= c->entries.size()) return ZKLC_ERR_INVALID_ARG;
    *out = c->entries[i];
    return ZKLC_OK;
}

extern "C" int32_t zklc_container_find(const zklc_container *c, uint32_t tag, zklc_container_entry *out) {
    if (!c || !out) return ZKLC_ERR_INVALID_ARG;
    for (const zklc_container_entry &e : c->entries)
        if (e.tag == tag) {
            *out = e;
            return ZKLC_OK;
        }
    return ZKLC_ERR_NOT_FOUND;
}

// the pages of the big sections are only needed until the circuit is on the GPU: give them back to the page cache
extern "C" void zklc_container_release_pages(const zklc_container *c) {
    if (c && c->map) (void)madvise(c->map, c->bytes, MADV_DONTNEED);
}

namespace {
struct circuit_view {
    const zklc_plonky2_params *params;
    const zklc_plonky2_gate *gates;
    const u64 *extra;
    u32 extra_words;
    const u64 *k_is, *constants, *sigmas;
};

// a section with exactly `count` elements of `elem` bytes
int32_t need(const zklc_container *c, u32 tag, u32 elem, u64 count, const void **p) {
    zklc_container_entry e;
    int32_t rc = zklc_container_find(c, tag, &e);
    if (rc == ZKLC_ERR_NOT_FOUND && count == 0) {      // an empty array may be left out
        *p = nullptr;
        return ZKLC_OK;
    }
    if (rc) return rc == ZKLC_ERR_NOT_FOUND ? ZKLC_ERR_FORMAT : rc;
    if (e.elem_bytes != elem || e.bytes != count * elem) return ZKLC_ERR_FORMAT;
    *p = e.data;
    return ZKLC_OK;
}

int32_t circuit_sections(const zklc_container *c, circuit_view *v) {
    const void *p;
    int32_t rc = need(c, ZKLC_SEC_PARAMS, sizeof(zklc_plonky2_params), 1, &p);
    if (rc) return rc;
    v->params = (const zklc_plonky2_params *)p;
    const zklc_plonky2_params &P = *v->params;
    if (P.degree_bits == 0 || P.degree_bits > 24 || P.num_gates == 0 || P.num_gates > 4096 || P.num_wires > 4096 ||
        P.num_routed_wires > P.num_wires || P.num_constants > 4096)
        return ZKLC_ERR_FORMAT;
    const u64 n = (u64)1 << P.degree_bits;
    if ((rc = need(c, ZKLC_SEC_GATES, sizeof(zklc_plonky2_gate), P.num_gates, &p))) return rc;
    v->gates = (const zklc_plonky2_gate *)p;
    zklc_container_entry ex;
    v->extra = nullptr;
    v->extra_words = 0;
    if (zklc_container_find(c, ZKLC_SEC_GATE_EXTRA, &ex) == ZKLC_OK) {
        if (ex.elem_bytes != 8 || ex.bytes / 8 > 0xffffffffu) return ZKLC_ERR_FORMAT;
        v->extra = (const u64 *)ex.data;
        v->extra_words = (u32)(ex.bytes / 8);
    }
    for (u32 g = 0; g < P.num_gates; g++)
        if (v->gates[g].extra_off > v->extra_words || v->gates[g].group_start > v->gates[g].group_end ||
            v->gates[g].group_end > P.num_gates || v->gates[g].selector_index >= P.num_selectors)
            return ZKLC_ERR_FORMAT;
    if ((rc = need(c, ZKLC_SEC_K_IS, 8, P.num_routed_wires, &p))) return rc;
    v->k_is = (const u64 *)p;
    if ((rc = need(c, ZKLC_SEC_CONSTANTS, 8, (u64)P.num_constants * n, &p))) return rc;
    v->constants = (const u64 *)p;
    if ((rc = need(c, ZKLC_SEC_SIGMAS, 8, (u64)P.num_routed_wires * n, &p))) return rc;
    v->sigmas = (const u64 *)p;
    return ZKLC_OK;
}

struct program_view {
    const zklc_witness_dims *d;
    const u32 *code, *input_slots, *wire_slot, *wire_index, *pi_slots;
    const int64_t *params;
};

int32_t program_sections(const zklc_container *c, program_view *v) {
    const void *p;
    int32_t rc = need(c, ZKLC_SEC_WP_DIMS, sizeof(zklc_witness_dims), 1, &p);
    if (rc) return rc;
    v->d = (const zklc_witness_dims *)p;
    const zklc_witness_dims &D = *v->d;
    if (!D.n_slots || !D.num_wires || !D.n_rows || (u64)D.num_wires * D.n_rows >> 32) return ZKLC_ERR_FORMAT;
    if ((rc = need(c, ZKLC_SEC_WP_CODE, 4, D.code_len, &p))) return rc;
    v->code = (const u32 *)p;
    if ((rc = need(c, ZKLC_SEC_WP_PARAMS, 8, D.n_params, &p))) return rc;
    v->params = (const int64_t *)p;
    if ((rc = need(c, ZKLC_SEC_WP_INPUT_SLOTS, 4, D.n_inputs, &p))) return rc;
    v->input_slots = (const u32 *)p;
    if ((rc = need(c, ZKLC_SEC_WP_WIRE_SLOT, 4, D.n_wire_entries, &p))) return rc;
    v->wire_slot = (const u32 *)p;
    if ((rc = need(c, ZKLC_SEC_WP_WIRE_INDEX, 4, D.n_wire_entries, &p))) return rc;
    v->wire_index = (const u32 *)p;
    if ((rc = need(c, ZKLC_SEC_WP_PI_SLOTS, 4, D.n_pi, &p))) return rc;
    v->pi_slots = (const u32 *)p;
    if (!v->code || !v->params) return ZKLC_ERR_FORMAT;
    // the interpreters index their value array by slot and the wire matrix by cell: a file is not trusted to be in range
    // (the instruction words themselves are checked by the create / run functions)
    const u32 cells = D.num_wires * D.n_rows;
    for (u64 k = 0; k < D.n_wire_entries; k++)
        if (v->wire_slot[k] >= D.n_slots || v->wire_index[k] >= cells) return ZKLC_ERR_FORMAT;
    for (u32 k = 0; k < D.n_inputs; k++)
        if (v->input_slots[k] >= D.n_slots) return ZKLC_ERR_FORMAT;
    for (u32 k = 0; k < D.n_pi; k++)
        if (v->pi_slots[k] >= D.n_slots) return ZKLC_ERR_FORMAT;
    // the instruction stream, structurally: [opcode, n_params, n_in, n_out, in slots.., out slots..]* must tile the code words exactly,
    // every slot must exist and the parameter blocks must lie inside the parameter array (the host interpreter indexes with these
    // numbers unchecked; the device scheduler checks them again).  What a parameter MEANS to its opcode is the writer's business:
    // the file comes from whoever built the circuit, and a wrong parameter is no worse than a wrong constant of the circuit itself.
    u64 ip = 0, pp = 0;
    while (ip < D.code_len) {
        if (ip + 4 > D.code_len) return ZKLC_ERR_FORMAT;
        const u64 np = v->code[ip + 1], ni = v->code[ip + 2], no = v->code[ip + 3];
        if (v->code[ip] > 255 || ip + 4 + ni + no > D.code_len || pp + np > D.n_params) return ZKLC_ERR_FORMAT;
        for (u64 k = 0; k < ni + no; k++)
            if (v->code[ip + 4 + k] >= D.n_slots) return ZKLC_ERR_FORMAT;
        ip += 4 + ni + no;
        pp += np;
    }
    return ZKLC_OK;
}
}  // namespace

extern "C" int32_t zklc_plonky2_container_params(const zklc_container *c, zklc_plonky2_params *params_out, zklc_witness_dims *dims_out) {
    if (!c) return ZKLC_ERR_INVALID_ARG;
    if (params_out) {
        circuit_view v;
        int32_t rc = circuit_sections(c, &v);
        if (rc) return rc;
        *params_out = *v.params;
    }
    if (dims_out) {
        const void *p;
        int32_t rc = need(c, ZKLC_SEC_WP_DIMS, sizeof(zklc_witness_dims), 1, &p);
        if (rc) return rc;
        *dims_out = *(const zklc_witness_dims *)p;
    }
    return ZKLC_OK;
}

extern "C" int32_t zklc_plonky2_circuit_create_from_container(zklc_ctx *ctx, const zklc_container *c, int32_t hasher,
                                                              zklc_plonky2_circuit **out) {
    if (!ctx || !c || !out) return ZKLC_ERR_INVALID_ARG;
    circuit_view v;
    int32_t rc = circuit_sections(c, &v);
    if (rc) return rc;
    zklc_plonky2_params P = *v.params;
    if (hasher >= 0) P.hasher = (u32)hasher;      // the circuit is the same under either Merkle hasher; the caller picks (wrap: BN128)
    return zklc_plonky2_circuit_create(ctx, &P, v.gates, v.extra, v.extra_words, v.k_is, v.constants, v.sigmas, out);
}

extern "C" int32_t zklc_plonky2_witness_program_create_from_container(zklc_ctx *ctx, const zklc_container *c, zklc_witness_program **out) {
    if (!ctx || !c || !out) return ZKLC_ERR_INVALID_ARG;
    program_view v;
    int32_t rc = program_sections(c, &v);
    if (rc) return rc;
    const zklc_witness_dims &D = *v.d;
    return zklc_plonky2_witness_program_create(ctx, v.code, D.code_len, v.params, D.n_params, D.n_slots, v.input_slots, D.n_inputs,
                                               v.wire_slot, v.wire_index, D.n_wire_entries, D.num_wires, D.n_rows, v.pi_slots, D.n_pi, out);
}

extern "C" int32_t zklc_plonky2_witness_run_from_container(const zklc_container *c, const uint64_t *input_values, uint32_t n_witnesses,
                                                           uint64_t *wires_out, uint64_t *pi_out, int32_t *status, char *err_out,
                                                           uint32_t threads) {
    if (!c || !wires_out || !status) return ZKLC_ERR_INVALID_ARG;
    program_view v;
    int32_t rc = program_sections(c, &v);
    if (rc) return rc;
    const zklc_witness_dims &D = *v.d;
    if (D.n_params == 0) return ZKLC_ERR_FORMAT;         // the interpreter reads one word past the last parameter block
    return zklc_plonky2_witness_run(v.code, D.code_len, v.params, D.n_slots, v.input_slots, D.n_inputs, input_values, n_witnesses,
                                    v.wire_slot, v.wire_index, D.n_wire_entries, D.num_wires, D.n_rows, wires_out, v.pi_slots, D.n_pi,
                                    pi_out, status, err_out, threads);
}

extern "C" int32_t zklc_plonky2_container_write(const char *path, const zklc_plonky2_params *params, const zklc_plonky2_gate *gates,
                                                const uint64_t *gate_extra, uint32_t gate_extra_words, const uint64_t *k_is,
                                                const uint64_t *constants, const uint64_t *sigmas, const zklc_witness_dims *dims,
                                                const uint32_t *code, const int64_t *wparams, const uint32_t *input_slots,
                                                const uint32_t *wire_slot, const uint32_t *wire_index, const uint32_t *pi_slots,
                                                const zklc_container_entry *extra_entries, uint32_t n_extra_entries) {
    if (!path || !params || !gates || !k_is || !constants || !sigmas || (gate_extra_words && !gate_extra) ||
        (n_extra_entries && !extra_entries) || params->degree_bits == 0 || params->degree_bits > 24)
        return ZKLC_ERR_INVALID_ARG;
    const u64 n = (u64)1 << params->degree_bits;
    std::vector<zklc_container_entry> e;
    e.push_back({ZKLC_SEC_PARAMS, (u32)sizeof(zklc_plonky2_params), params, sizeof(zklc_plonky2_params)});
    e.push_back({ZKLC_SEC_GATES, (u32)sizeof(zklc_plonky2_gate), gates, (u64)params->num_gates * sizeof(zklc_plonky2_gate)});
    if (gate_extra_words) e.push_back({ZKLC_SEC_GATE_EXTRA, 8, gate_extra, (u64)gate_extra_words * 8});
    e.push_back({ZKLC_SEC_K_IS, 8, k_is, (u64)params->num_routed_wires * 8});
    e.push_back({ZKLC_SEC_CONSTANTS, 8, constants, (u64)params->num_constants * n * 8});
    e.push_back({ZKLC_SEC_SIGMAS, 8, sigmas, (u64)params->num_routed_wires * n * 8});
    if (dims) {
        if (!code || !wparams || (dims->n_inputs && !input_slots) || (dims->n_wire_entries && (!wire_slot || !wire_index)) ||
            (dims->n_pi && !pi_slots))
            return ZKLC_ERR_INVALID_ARG;
        e.push_back({ZKLC_SEC_WP_DIMS, (u32)sizeof(zklc_witness_dims), dims, sizeof(zklc_witness_dims)});
        e.push_back({ZKLC_SEC_WP_CODE, 4, code, dims->code_len * 4});
        e.push_back({ZKLC_SEC_WP_PARAMS, 8, wparams, dims->n_params * 8});
        if (dims->n_inputs) e.push_back({ZKLC_SEC_WP_INPUT_SLOTS, 4, input_slots, (u64)dims->n_inputs * 4});
        if (dims->n_wire_entries) {
            e.push_back({ZKLC_SEC_WP_WIRE_SLOT, 4, wire_slot, dims->n_wire_entries * 4});
            e.push_back({ZKLC_SEC_WP_WIRE_INDEX, 4, wire_index, dims->n_wire_entries * 4});
        }
        if (dims->n_pi) e.push_back({ZKLC_SEC_WP_PI_SLOTS, 4, pi_slots, (u64)dims->n_pi * 4});
    }
    for (u32 i = 0; i < n_extra_entries; i++) {
        if (extra_entries[i].tag < ZKLC_SEC_HOST_FIRST) return ZKLC_ERR_INVALID_ARG;     // the tags below are the library's
        e.push_back(extra_entries[i]);
    }
    return zklc_container_write(path, e.data(), (u32)e.size());
}
