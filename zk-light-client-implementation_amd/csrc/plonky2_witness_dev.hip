// Witness generation on the GPU: the generators that run inside `CircuitData::prove` of the reference
//   crypto/plonky2_ed25519/src/gadgets/nonnative.rs:447-705, gadgets/curve.rs:327-370, plonky2_ecdsa/src/gadgets/biguint.rs:417-470,
//   crypto/plonky2_u32/src/gates/*.rs `generators()`, plonky2's own gate generators (un-vendored)
// executed by the device for a BATCH of witnesses, so that the wire matrices (490 MB per Ed25519 signature) are produced in HBM
// in the prover's layout and never cross PCIe, and the host cores are out of the per-signature path.
//
// The builder's instruction stream (one instruction per generator; semantics in plonky2_witness_ops.h, shared with the host
// interpreter) is a DAG over value slots (a slot = a copy class).  At program creation the instructions are levelled -- level =
// 1 + the highest level of an input, a slot being available from its FIRST writer on -- and sorted by (level, opcode): the
// Ed25519 circuit's 1.13 M instructions have a critical path of 3 661 levels (SHA-512, then 1 272 non-native multiplications, 318
// inversions, 700 additions / subtractions of the scalar multiplications).  A level is executed by one lane per
// (instruction, witness): consecutive lanes = the W witnesses of one instruction, so control flow is uniform over an instruction
// and the slot values of a batch, stored as val[slot][W], are read and written with W-wide coalesced accesses.  Runs of small
// levels (<= 1024 lanes) go to ONE workgroup that steps through them with a barrier per level; a large level is one launch.
// Every slot access is an agent-scope atomic (a compare-and-swap against the "unassigned" sentinel for writes: a second writer
// with a different value is the reference's "copy constraint violated", the failure an invalid signature produces), so the
// values are coherent across workgroups, XCDs and launches without any further fence.
#include <algorithm>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "plonky2_witness_ops.h"
#include "zklc_internal.h"

#define WIT_SENT 0xFFFFFFFFFFFFFFFFULL      // not a field element: the slot has no value yet
#define WIT_STEP_THREADS 1024u               // the single-workgroup stepping kernel
#define WIT_SMALL 4096u                     // it takes the levels of at most this many lanes

struct wit_sched {
    u32 ip, pp;                             // offsets of the instruction in code[] / params[]
};

struct zklc_witness_program {
    zklc_ctx *ctx = nullptr;
    int device = 0;
    u32 *d_code = nullptr;
    int64_t *d_params = nullptr;
    u32 *d_input_slots = nullptr, *d_wire_slot = nullptr, *d_wire_index = nullptr, *d_pi_slots = nullptr;
    // the schedule padded for a lane group of Wp = 2^k witnesses: every opcode group of a level starts on a wavefront boundary,
    // so a wave never executes two different generators one after the other (built on first use)
    struct padded {
        wit_sched *d_sched = nullptr;
        u32 *d_level_start = nullptr;
        std::vector<u32> level_start, level_heavy;      // in padded instruction slots; heavy = trailing slots of the level
    } per_wp[7];
    std::vector<wit_sched> sched;           // unpadded, sorted by (level, heavy, opcode)
    u64 n_wire_entries = 0, n_instr = 0;
    u32 n_slots = 0, n_inputs = 0, n_pi = 0, num_wires = 0, n_rows = 0, n_levels = 0;
    std::vector<u32> level_start;           // n_levels + 1 (unpadded)
    std::vector<uint8_t> sched_ops;         // opcode of every scheduled instruction
    std::vector<u32> level_heavy;           // per level: its last level_heavy[l] instructions need the generic big-integer kernel
    u64 n_heavy = 0;
    // per-run buffers, grown on demand
    u64 *d_val = nullptr, *d_inputs = nullptr, *d_pis = nullptr;
    unsigned long long *d_err = nullptr;
    u32 cap_w = 0;
    // page-locked staging of a batch's inputs, public inputs and error words: hipMemcpyAsync to / from pageable memory waits for
    // the stream inside the call, actively (csrc/plonky2_prover.hip `p2_pin`); from here the thread sleeps in zklc_stream_wait
    uint8_t *h_pin = nullptr;
    size_t h_pin_bytes = 0;
    std::vector<void *> allocs;
};

struct wit_args {
    const u32 *code;
    const int64_t *params;
    const wit_sched *sched;
    const u32 *level_start;
    u64 *val;
    unsigned long long *err;
    u32 W, Wp;                              // witnesses in the batch; lanes per instruction (a power of two >= W)
    unsigned long long *trace;               // optional: wall clock at the end of every level of the stepping kernel
};

struct wit_dev_io {
    u64 *val;
    const u32 *is, *os;
    unsigned long long *err;
    u32 W, w, no, k, pc;
    u32 bad;                                // a slot already held another value: reported once, after the instruction
    u32 single;                             // every output slot has this instruction as its only writer: plain (agent-scope) stores
    __device__ u64 in(u32 i) const {
        return __hip_atomic_load(&val[(size_t)is[i] * W + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ bool fail(int code) {
        atomicCAS(&err[w], 0ULL, ((unsigned long long)code << 32) | (unsigned long long)(pc + 1));
        return false;
    }
    __device__ bool put(u32 slot, u64 v) {
        if (v >= GL_P) v -= GL_P;
        unsigned long long *p = (unsigned long long *)&val[(size_t)slot * W + w];
        if (single) {
            __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        // no branch on the result: the compare-and-swaps of an instruction's outputs (up to 35 for a u32 multiply-add) stay in
        // flight together instead of paying one memory round trip each
        unsigned long long old = atomicCAS(p, (unsigned long long)WIT_SENT, (unsigned long long)v);
        bad |= (old != WIT_SENT && old != v) ? 1u : 0u;
        return true;
    }
    __device__ bool out_at(u32 idx, u64 v) {   // out-of-order outputs (PoseidonGate rows); the instruction then counts as complete
        k = no;
        return idx < no ? put(os[idx], v) : fail(WIT_ERR_OUT_COUNT);
    }
    __device__ bool out(u64 v) {
        if (k >= no) {
            k++;
            return true;                    // reported as an output-count mismatch after the instruction
        }
        return put(os[k++], v);
    }
};

#define WIT_NOOP 0xFFFFFFFFu
template <bool HEAVY>
__device__ __noinline__ void wit_run_one(const wit_args &a, u32 slot, u32 w) {
    wit_sched s = a.sched[slot];
    if (s.ip == WIT_NOOP) return;                                                            // alignment padding
    if (__hip_atomic_load(&a.err[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // this witness already failed
    const u32 *c = a.code + s.ip;
    int op = (int)c[0];
    u32 np = c[1], ni = c[2], no = c[3];
    wit_dev_io io = {a.val, c + 4, c + 4 + ni, a.err, a.W, w, no, 0, slot, 0, s.pp >> 31};
    bool ok = wit_exec<HEAVY>(op, a.params + (s.pp & 0x7FFFFFFFu), np, ni, no, io);
    if (io.bad) io.fail(WIT_ERR_COPY);
    else if (ok && io.k != no) io.fail(WIT_ERR_OUT_COUNT);
}

// schedule slots [first, first + count) of one level; lane t -> slot first + t / Wp, witness t % Wp
template <bool HEAVY>
__global__ void __launch_bounds__(256) wit_level_kernel(wit_args a, u32 first, u32 count) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * a.Wp || t % a.Wp >= a.W) return;
    wit_run_one<HEAVY>(a, first + t / a.Wp, t % a.Wp);
}

// levels [l0, l1) without heavy instructions, each of at most WIT_SMALL lanes, stepped through by one workgroup
__global__ void __launch_bounds__(WIT_STEP_THREADS) wit_levels_small_kernel(wit_args a, u32 l0, u32 l1) {
    for (u32 l = l0; l < l1; l++) {
        u32 first = a.level_start[l], lanes = (a.level_start[l + 1] - first) * a.Wp;
        for (u32 t = threadIdx.x; t < lanes; t += WIT_STEP_THREADS)
            if (t % a.Wp < a.W) wit_run_one<false>(a, first + t / a.Wp, t % a.Wp);
        __syncthreads();
        if (a.trace && threadIdx.x == 0) a.trace[l] = wall_clock64();
    }
}

__global__ void wit_set_inputs_kernel(u64 *val, const u32 *slots, const u64 *values, u32 n_inputs, u32 W, unsigned long long *err) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_inputs * W) return;
    u32 i = t / W, w = t % W;
    u64 v = values[(size_t)w * n_inputs + i] % GL_P;
    unsigned long long *p = (unsigned long long *)&val[(size_t)slots[i] * W + w];
    unsigned long long old = atomicCAS(p, (unsigned long long)WIT_SENT, (unsigned long long)v);
    if (old != WIT_SENT && old != v) atomicCAS(&err[w], 0ULL, ((unsigned long long)WIT_ERR_COPY << 32));
}

// wires[w][wire_index[k]] = val[wire_slot[k]][w] for the assigned slots.  A workgroup stages 256 entries x W witnesses through LDS:
// reads are W-wide lines of val, writes are runs of consecutive cells of one wire matrix (the list is sorted by address).
#define WIT_SC_TILE 256
__global__ void __launch_bounds__(256) wit_scatter_kernel(const u64 *val, const u32 *wire_slot, const u32 *wire_index, u64 n_entries, u32 W,
                                                          u64 *wires, u64 matrix_words) {
    extern __shared__ u64 wit_sh[];          // [W][WIT_SC_TILE + 1]
    u64 k0 = (u64)blockIdx.x * WIT_SC_TILE;
    u32 n = (u32)(n_entries - k0 < WIT_SC_TILE ? n_entries - k0 : WIT_SC_TILE);
    for (u32 t = threadIdx.x; t < n * W; t += blockDim.x) {
        u32 k = t / W, w = t % W;
        wit_sh[(size_t)w * (WIT_SC_TILE + 1) + k] = val[(size_t)wire_slot[k0 + k] * W + w];
    }
    __syncthreads();
    for (u32 t = threadIdx.x; t < n * W; t += blockDim.x) {
        u32 w = t / n, k = t % n;
        u64 v = wit_sh[(size_t)w * (WIT_SC_TILE + 1) + k];
        if (v != WIT_SENT) wires[(size_t)w * matrix_words + wire_index[k0 + k]] = v;   // an unassigned class keeps the buffer's zero
    }
}

__global__ void wit_public_inputs_kernel(const u64 *val, const u32 *pi_slots, u32 n_pi, u32 W, u64 *out, unsigned long long *err) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pi * W) return;
    u32 k = t / W, w = t % W;
    u64 v = val[(size_t)pi_slots[k] * W + w];
    if (v == WIT_SENT) {
        atomicCAS(&err[w], 0ULL, ((unsigned long long)WIT_ERR_PI << 32) | (unsigned long long)(k + 1));
        v = 0;
    }
    out[(size_t)w * n_pi + k] = v;
}

static int32_t wit_alloc(zklc_witness_program *p, void **ptr, size_t bytes) {
    zklc_ctx *ctx = p->ctx;
    ZKLC_HIP(ctx, hipMalloc(ptr, bytes ? bytes : 8));
    p->allocs.push_back(*ptr);
    return ZKLC_OK;
}
#define WIT_ALLOC(p, ptr, bytes)                                      \
    do {                                                              \
        int32_t rc__ = wit_alloc((p), (void **)&(ptr), (bytes));      \
        if (rc__) return rc__;                                        \
    } while (0)

extern "C" void zklc_plonky2_witness_program_destroy(zklc_witness_program *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipDeviceSynchronize();
    for (void *q : p->allocs) (void)hipFree(q);
    if (p->h_pin) (void)hipHostFree(p->h_pin);
    delete p;
}

static int32_t wit_create(zklc_ctx *ctx, zklc_witness_program *p, const uint32_t *code, uint64_t code_len, const int64_t *params,
                          uint64_t n_params, uint32_t n_slots, const uint32_t *input_slots, uint32_t n_inputs, const uint32_t *wire_slot,
                          const uint32_t *wire_index, uint64_t n_wire_entries, uint32_t num_wires, uint32_t n_rows,
                          const uint32_t *pi_slots, uint32_t n_pi) {
    p->ctx = ctx;
    p->device = ctx->device;
    p->n_slots = n_slots;
    p->n_inputs = n_inputs;
    p->n_pi = n_pi;
    p->num_wires = num_wires;
    p->n_rows = n_rows;
    p->n_wire_entries = n_wire_entries;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    // ---- levels: a slot is available from its first writer on; an instruction runs one level after its last input
    const u32 INF = 0xFFFFFFFFu;
    std::vector<u32> avail(n_slots, INF);
    std::vector<uint8_t> writers(n_slots, 0);          // writers per slot, saturating; a circuit input counts as one
    for (u32 i = 0; i < n_inputs; i++) {
        if (input_slots[i] >= n_slots) return ZKLC_ERR_INVALID_ARG;
        avail[input_slots[i]] = 0;
        writers[input_slots[i]] = 1;
    }
    struct rec {
        u32 level, op, ip, pp, heavy;
    };
    std::vector<rec> recs;
    u64 ip = 0, pp = 0;
    u32 max_level = 0;
    while (ip < code_len) {
        if (ip + 4 > code_len) return ZKLC_ERR_INVALID_ARG;
        u32 op = code[ip], np = code[ip + 1], ni = code[ip + 2], no = code[ip + 3];
        if (ip + 4 + (u64)ni + no > code_len || pp + np > n_params) return ZKLC_ERR_INVALID_ARG;
        const u32 *is = code + ip + 4, *os = is + ni;
        u32 m = 0;
        for (u32 i = 0; i < ni; i++) {
            if (is[i] >= n_slots) return ZKLC_ERR_INVALID_ARG;
            if (avail[is[i]] == INF) {
                ctx->last_err = "witness program: an instruction reads a slot no earlier instruction writes";
                return ZKLC_ERR_INVALID_ARG;
            }
            if (avail[is[i]] > m) m = avail[is[i]];
        }
        m++;
        for (u32 i = 0; i < no; i++) {
            if (os[i] >= n_slots) return ZKLC_ERR_INVALID_ARG;
            if (m < avail[os[i]]) avail[os[i]] = m;
            if (writers[os[i]] < 255) writers[os[i]]++;
        }
        recs.push_back({m, op, (u32)ip, (u32)pp, wit_is_heavy((int)op, params + pp, ni) ? 1u : 0u});
        if (m > max_level) max_level = m;
        ip += 4 + (u64)ni + no;
        pp += np;
    }
    if (ip >> 32 || pp >> 32) return ZKLC_ERR_INVALID_ARG;
    std::stable_sort(recs.begin(), recs.end(), [](const rec &x, const rec &y) {
        return x.level != y.level ? x.level < y.level : x.heavy != y.heavy ? x.heavy < y.heavy : x.op < y.op;
    });
    p->n_instr = recs.size();
    p->n_levels = max_level;
    p->level_start.assign(max_level + 1, 0);
    p->level_heavy.assign(max_level, 0);
    std::vector<wit_sched> sched(recs.size());
    {
        size_t k = 0;
        for (u32 l = 1; l <= max_level; l++) {
            p->level_start[l - 1] = (u32)k;
            while (k < recs.size() && recs[k].level == l) {
                p->level_heavy[l - 1] += recs[k].heavy;
                p->n_heavy += recs[k].heavy;
                k++;
            }
        }
        p->level_start[max_level] = (u32)recs.size();
        p->sched_ops.resize(recs.size());
        for (size_t i = 0; i < recs.size(); i++) {
            const u32 *c = code + recs[i].ip;
            bool single = true;                        // no other instruction (or circuit input) writes any of its outputs
            for (u32 k = 0; k < c[3]; k++) single = single && writers[c[4 + c[2] + k]] == 1;
            if (recs[i].pp >> 31) return ZKLC_ERR_INVALID_ARG;
            sched[i] = {recs[i].ip, recs[i].pp | (single ? 0x80000000u : 0u)};
            p->sched_ops[i] = (uint8_t)recs[i].op;
        }
        p->sched = sched;
    }
    hipStream_t st = ctx->stream;
    WIT_ALLOC(p, p->d_code, code_len * 4);
    WIT_ALLOC(p, p->d_params, (n_params + 1) * 8);
    WIT_ALLOC(p, p->d_input_slots, (size_t)n_inputs * 4);
    WIT_ALLOC(p, p->d_wire_slot, n_wire_entries * 4);
    WIT_ALLOC(p, p->d_wire_index, n_wire_entries * 4);
    WIT_ALLOC(p, p->d_pi_slots, (size_t)n_pi * 4);
    ZKLC_HIP(ctx, hipMemcpyAsync(p->d_code, code, code_len * 4, hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, hipMemcpyAsync(p->d_params, params, n_params * 8, hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, hipMemcpyAsync(p->d_input_slots, input_slots, (size_t)n_inputs * 4, hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, hipMemcpyAsync(p->d_wire_slot, wire_slot, n_wire_entries * 4, hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, hipMemcpyAsync(p->d_wire_index, wire_index, n_wire_entries * 4, hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, hipMemcpyAsync(p->d_pi_slots, pi_slots, (size_t)n_pi * 4, hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, zklc_stream_wait(st));   // the sources are the caller's (and this function's) host buffers
    return ZKLC_OK;
}

extern "C" int32_t zklc_plonky2_witness_program_create(zklc_ctx *ctx, const uint32_t *code, uint64_t code_len, const int64_t *params,
                                                       uint64_t n_params, uint32_t n_slots, const uint32_t *input_slots,
                                                       uint32_t n_inputs, const uint32_t *wire_slot, const uint32_t *wire_index,
                                                       uint64_t n_wire_entries, uint32_t num_wires, uint32_t n_rows,
                                                       const uint32_t *pi_slots, uint32_t n_pi, zklc_witness_program **out) {
    if (!ctx || !code || !params || !out || (n_inputs && !input_slots) || (n_wire_entries && (!wire_slot || !wire_index)) ||
        (n_pi && !pi_slots) || !n_slots || !num_wires || !n_rows)
        return ZKLC_ERR_INVALID_ARG;
    zklc_witness_program *p = new (std::nothrow) zklc_witness_program();
    if (!p) return ZKLC_ERR_OOM;
    int32_t rc = wit_create(ctx, p, code, code_len, params, n_params, n_slots, input_slots, n_inputs, wire_slot, wire_index, n_wire_entries,
                            num_wires, n_rows, pi_slots, n_pi);
    if (rc) {
        zklc_plonky2_witness_program_destroy(p);
        return rc;
    }
    *out = p;
    return ZKLC_OK;
}

static u32 wit_wp_index(u32 W) {
    u32 k = 0;
    while ((1u << k) < W) k++;
    return k;
}

// the schedule for lane groups of Wp witnesses: within a level every (class, opcode) group starts at a multiple of 64 / Wp
// slots, i.e. on a wavefront boundary; padding slots hold WIT_NOOP
static int32_t wit_padded(zklc_witness_program *p, u32 k, hipStream_t st) {
    zklc_ctx *ctx = p->ctx;
    zklc_witness_program::padded &P = p->per_wp[k];
    if (P.d_sched) return ZKLC_OK;
    const u32 G = 64u >> k;                  // slots per wavefront
    std::vector<wit_sched> out;
    out.reserve(p->sched.size() + p->sched.size() / 4);
    P.level_start.assign(p->n_levels + 1, 0);
    P.level_heavy.assign(p->n_levels, 0);
    for (u32 l = 0; l < p->n_levels; l++) {
        P.level_start[l] = (u32)out.size();
        u32 b = p->level_start[l], e = p->level_start[l + 1], light_end = e - p->level_heavy[l];
        auto emit_groups = [&](u32 from, u32 to) {
            for (u32 i = from; i < to;) {
                u32 j = i;
                while (j < to && p->sched_ops[j] == p->sched_ops[i]) j++;
                while (out.size() % G) out.push_back({WIT_NOOP, 0});
                for (u32 q = i; q < j; q++) out.push_back(p->sched[q]);
                i = j;
            }
        };
        emit_groups(b, light_end);
        while (out.size() % G) out.push_back({WIT_NOOP, 0});
        u32 heavy_begin = (u32)out.size();
        emit_groups(light_end, e);
        P.level_heavy[l] = (u32)out.size() - heavy_begin;
    }
    P.level_start[p->n_levels] = (u32)out.size();
    WIT_ALLOC(p, P.d_sched, out.size() * sizeof(wit_sched));
    WIT_ALLOC(p, P.d_level_start, P.level_start.size() * 4);
    ZKLC_HIP(ctx, hipMemcpyAsync(P.d_sched, out.data(), out.size() * sizeof(wit_sched), hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, hipMemcpyAsync(P.d_level_start, P.level_start.data(), P.level_start.size() * 4, hipMemcpyHostToDevice, st));
    ZKLC_HIP(ctx, zklc_stream_wait(st));
    return ZKLC_OK;
}

// the launches of a batch: f(kind, x, y) with kind 0 = stepping kernel over levels [x, y), 1 = light kernel over schedule slots
// [x, x + y), 2 = heavy kernel over slots [x, x + y)
template <class F>
static void wit_plan(const zklc_witness_program::padded &P, u32 n_levels, u32 Wp, F f) {
    for (u32 l = 0; l < n_levels;) {
        auto small = [&](u32 k) { return !P.level_heavy[k] && (u64)(P.level_start[k + 1] - P.level_start[k]) * Wp <= WIT_SMALL; };
        if (small(l)) {
            u32 e = l + 1;
            while (e < n_levels && small(e)) e++;
            f(0, l, e);
            l = e;
            continue;
        }
        u32 first = P.level_start[l], count = P.level_start[l + 1] - first, heavy = P.level_heavy[l];
        if (count > heavy) f(1, first, count - heavy);
        if (heavy) f(2, first + count - heavy, heavy);
        l++;
    }
}

// instructions, levels, and the number of kernel launches a batch of `n_witnesses` takes
extern "C" int32_t zklc_plonky2_witness_program_info(zklc_witness_program *p, uint32_t n_witnesses, uint64_t *n_instr, uint32_t *n_levels,
                                                     uint32_t *n_launches) {
    if (!p || !n_witnesses) return ZKLC_ERR_INVALID_ARG;
    if (n_instr) *n_instr = p->n_instr;
    if (n_levels) *n_levels = p->n_levels;
    if (n_launches) {
        u32 launches = 0;
        u32 k = wit_wp_index(n_witnesses);
        if (k > 6) return ZKLC_ERR_INVALID_ARG;
        int32_t rc = wit_padded(p, k, p->ctx->stream);
        if (rc) return rc;
        wit_plan(p->per_wp[k], p->n_levels, 1u << k, [&](int, u32, u32) { launches++; });
        *n_launches = launches;
    }
    return ZKLC_OK;
}

// Runs the program for n_witnesses partial witnesses (input_values: host, n_witnesses x n_inputs) and writes the wire matrices into
// d_wires (DEVICE, n_witnesses x num_wires x n_rows; only the circuit's wire cells are written, the caller zero-fills the buffer
// once) on `stream`.  pi_out (host, n_witnesses x n_pi), status (host, 0 ok / 1 failed) and err_out (host, optional, 200 bytes per
// witness) are valid on return: the call synchronises the stream.
extern "C" int32_t zklc_plonky2_witness_run_dev(zklc_ctx *ctx, void *stream, zklc_witness_program *p, const uint64_t *input_values,
                                                uint32_t n_witnesses, uint64_t *d_wires, uint64_t *pi_out, int32_t *status,
                                                char *err_out) {
    if (!ctx || !p || p->ctx != ctx || !d_wires || !status || !n_witnesses || n_witnesses > 64 || (p->n_inputs && !input_values) ||
        (p->n_pi && !pi_out))
        return ZKLC_ERR_INVALID_ARG;
    ZKLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = zklc_pick_stream(ctx, stream);
    const u32 W = n_witnesses;
    if (p->cap_w < W) {
        // (re)allocate the batch buffers for W witnesses
        for (void **q : {(void **)&p->d_val, (void **)&p->d_inputs, (void **)&p->d_pis, (void **)&p->d_err})
            if (*q) {
                ZKLC_HIP(ctx, zklc_stream_wait(st));
                ZKLC_HIP(ctx, hipFree(*q));
                p->allocs.erase(std::remove(p->allocs.begin(), p->allocs.end(), *q), p->allocs.end());
                *q = nullptr;
            }
        p->cap_w = 0;
        WIT_ALLOC(p, p->d_val, (size_t)p->n_slots * W * 8);
        WIT_ALLOC(p, p->d_inputs, (size_t)p->n_inputs * W * 8);
        WIT_ALLOC(p, p->d_pis, (size_t)p->n_pi * W * 8);
        WIT_ALLOC(p, p->d_err, (size_t)64 * 8);
        p->cap_w = W;
    }
    const size_t in_bytes = (size_t)p->n_inputs * W * 8, pi_bytes = (size_t)p->n_pi * W * 8;
    const size_t in_off = 0, pi_off = (in_bytes + 63) & ~(size_t)63, err_off = pi_off + ((pi_bytes + 63) & ~(size_t)63);
    if (p->h_pin_bytes < err_off + 64 * 8) {
        if (p->h_pin) (void)hipHostFree(p->h_pin);      // no transfer of this program is in flight between two runs
        p->h_pin = nullptr;
        p->h_pin_bytes = 0;
        ZKLC_HIP(ctx, hipHostMalloc((void **)&p->h_pin, err_off + 64 * 8, hipHostMallocDefault));
        p->h_pin_bytes = err_off + 64 * 8;
    }
    ZKLC_HIP(ctx, hipMemsetAsync(p->d_val, 0xFF, (size_t)p->n_slots * W * 8, st));
    ZKLC_HIP(ctx, hipMemsetAsync(p->d_err, 0, 64 * 8, st));
    if (p->n_inputs) {
        memcpy(p->h_pin + in_off, input_values, in_bytes);
        ZKLC_HIP(ctx, hipMemcpyAsync(p->d_inputs, p->h_pin + in_off, in_bytes, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(wit_set_inputs_kernel, dim3((p->n_inputs * W + 255) / 256), dim3(256), 0, st, p->d_val,
                           (const u32 *)p->d_input_slots, (const u64 *)p->d_inputs, p->n_inputs, W, p->d_err);
    }
    const u32 kwp = wit_wp_index(W), Wp = 1u << kwp;
    {
        int32_t rc = wit_padded(p, kwp, st);
        if (rc) return rc;
    }
    const zklc_witness_program::padded &PS = p->per_wp[kwp];
    wit_args a = {p->d_code, p->d_params, PS.d_sched, PS.d_level_start, p->d_val, p->d_err, W, Wp, nullptr};
    unsigned long long *d_trace = nullptr;
    if (getenv("ZKLC_WIT_TRACE")) {
        ZKLC_HIP(ctx, hipMalloc((void **)&d_trace, (size_t)(p->n_levels + 1) * 8));
        ZKLC_HIP(ctx, hipMemsetAsync(d_trace, 0, (size_t)(p->n_levels + 1) * 8, st));
        a.trace = d_trace;
    }
    wit_plan(PS, p->n_levels, Wp, [&](int kind, u32 x, u32 y) {
        if (kind == 0)
            hipLaunchKernelGGL(wit_levels_small_kernel, dim3(1), dim3(WIT_STEP_THREADS), 0, st, a, x, y);
        else if (kind == 1)
            hipLaunchKernelGGL(wit_level_kernel<false>, dim3(((u64)y * Wp + 255) / 256), dim3(256), 0, st, a, x, y);
        else
            hipLaunchKernelGGL(wit_level_kernel<true>, dim3(((u64)y * Wp + 255) / 256), dim3(256), 0, st, a, x, y);
    });
    ZKLC_HIP(ctx, hipGetLastError());
    if (p->n_wire_entries)
        hipLaunchKernelGGL(wit_scatter_kernel, dim3((p->n_wire_entries + WIT_SC_TILE - 1) / WIT_SC_TILE), dim3(256),
                           (size_t)W * (WIT_SC_TILE + 1) * 8, st, (const u64 *)p->d_val, (const u32 *)p->d_wire_slot,
                           (const u32 *)p->d_wire_index, p->n_wire_entries, W, d_wires, (u64)p->num_wires * p->n_rows);
    if (p->n_pi)
        hipLaunchKernelGGL(wit_public_inputs_kernel, dim3((p->n_pi * W + 255) / 256), dim3(256), 0, st, (const u64 *)p->d_val,
                           (const u32 *)p->d_pi_slots, p->n_pi, W, p->d_pis, p->d_err);
    ZKLC_HIP(ctx, hipGetLastError());
    unsigned long long errs[64];
    // settled read-backs (zklc_internal.h): the copies are enqueued when the ~2 300 kernels of the batch have ended, not behind them
    if (zklc_settled_copies()) ZKLC_HIP(ctx, zklc_stream_wait(st));
    if (p->n_pi) ZKLC_HIP(ctx, hipMemcpyAsync(p->h_pin + pi_off, p->d_pis, pi_bytes, hipMemcpyDeviceToHost, st));
    ZKLC_HIP(ctx, hipMemcpyAsync(p->h_pin + err_off, p->d_err, 64 * 8, hipMemcpyDeviceToHost, st));
    ZKLC_HIP(ctx, zklc_stream_wait(st));
    if (p->n_pi) memcpy(pi_out, p->h_pin + pi_off, pi_bytes);
    memcpy(errs, p->h_pin + err_off, 64 * 8);
    if (d_trace) {
        // debugging aid: the slowest levels of the stepping kernel with their opcode mix (ticks of the 100 MHz wall clock)
        std::vector<unsigned long long> tr(p->n_levels + 1);
        ZKLC_HIP(ctx, hipMemcpy(tr.data(), d_trace, tr.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_trace);
        std::vector<std::pair<unsigned long long, u32>> d;
        unsigned long long total = 0;
        for (u32 l = 1; l < p->n_levels; l++)
            if (tr[l] && tr[l - 1] && tr[l] > tr[l - 1]) {
                d.push_back({tr[l] - tr[l - 1], l});
                total += tr[l] - tr[l - 1];
            }
        std::sort(d.rbegin(), d.rend());
        fprintf(stderr, "[zklc] witness trace: %zu stepped levels, %.1f ms in total (100 MHz ticks)\n", d.size(), total / 1e5);
        std::vector<u32> code_h;
        for (size_t k = 0; k < d.size() && k < 25; k++) {
            u32 l = d[k].second;
            fprintf(stderr, "   level %u: %.1f us, %u instructions, ops:", l, d[k].first / 100.0, p->level_start[l + 1] - p->level_start[l]);
            u32 cnt[32] = {};
            for (u32 i = p->level_start[l]; i < p->level_start[l + 1]; i++) cnt[p->sched_ops[i] & 31]++;
            for (int o = 0; o < 32; o++)
                if (cnt[o]) fprintf(stderr, " %d:%u", o, cnt[o]);
            fprintf(stderr, "\n");
        }
        if (const char *path = getenv("ZKLC_WIT_TRACE_FILE")) {
            if (FILE *f = fopen(path, "w")) {
                for (auto &x : d) {
                    u32 l = x.second, cnt[32] = {};
                    for (u32 i = p->level_start[l]; i < p->level_start[l + 1]; i++) cnt[p->sched_ops[i] & 31]++;
                    fprintf(f, "%u,%llu", l, x.first);
                    for (int o = 0; o < 32; o++) fprintf(f, ",%u", cnt[o]);
                    fprintf(f, "\n");
                }
                fclose(f);
            }
        }
        // histogram of level durations
        u32 hist[8] = {};
        for (auto &x : d) {
            double us = x.first / 100.0;
            hist[us < 10 ? 0 : us < 25 ? 1 : us < 50 ? 2 : us < 100 ? 3 : us < 200 ? 4 : us < 400 ? 5 : us < 1000 ? 6 : 7]++;
        }
        fprintf(stderr, "   levels by duration  <10us %u  <25 %u  <50 %u  <100 %u  <200 %u  <400 %u  <1ms %u  more %u\n", hist[0], hist[1], hist[2],
                hist[3], hist[4], hist[5], hist[6], hist[7]);
    }
    for (u32 w = 0; w < W; w++) {
        status[w] = errs[w] ? 1 : 0;
        if (err_out) {
            char *e = err_out + (size_t)w * 200;
            if (errs[w])
                snprintf(e, 200, "%s at scheduled instruction %llu", wit_strerror((int)(errs[w] >> 32)), errs[w] & 0xFFFFFFFFULL);
            else
                e[0] = 0;
        }
    }
    return ZKLC_OK;
}
