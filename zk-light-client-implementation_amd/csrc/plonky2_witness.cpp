// Native witness generation for circuits built by zklc_amd.plonky2.CircuitBuilder (host code, plain C++).
//
// Replaces the witness generators that run inside `CircuitData::prove` of the reference:
//   crypto/plonky2_ed25519/src/gadgets/nonnative.rs:447-705  NonNative{Addition,Subtraction,Multiplication,Inverse}Generator
//   crypto/plonky2_ed25519/src/gadgets/curve.rs:327-370      CurvePointDecompressionGenerator
//   crypto/plonky2_ecdsa/src/gadgets/biguint.rs:417-470      BigUintDivRemGenerator
//   crypto/plonky2_u32/src/gates/*.rs `generators()`          U32 arithmetic / add-many / subtraction / range-check /
//                                                             comparison gate generators
//   plonky2 (un-vendored) ArithmeticGate / BaseSumGate / RandomAccessGate / PoseidonGate / EqualityGenerator
// The builder records one instruction per generator (opcode, parameters, input slots, output slots; a slot = one copy
// class of the circuit); this interpreter executes the program for a batch of partial witnesses, one host thread per
// witness, and scatters the slot values into the poly-major wire matrix the prover takes.  Writing a slot twice with
// different values is the reference's "copy constraint violated" failure (e.g. an invalid signature) -> error.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <map>
#include <memory>
#include <tuple>
#include <mutex>
#include <thread>
#include <vector>
#include "plonky2_witness_ops.h"

static const u64 GLP = GL_P;

struct Runner {
    std::vector<u64> val;
    std::vector<u32> epoch;
    u32 cur = 0;
    char err[200];
    bool failed = false;
    std::mutex err_mutex;

    bool set(u32 slot, u64 v, u64 pc) {
        if (epoch[slot] == cur) {
            if (val[slot] != v) {
                std::lock_guard<std::mutex> lk(err_mutex);
                if (!failed)
                    snprintf(err, sizeof(err), "copy constraint violated at instruction %llu (slot %u: %llu != %llu)",
                             (unsigned long long)pc, slot, (unsigned long long)val[slot], (unsigned long long)v);
                failed = true;
                return false;
            }
            return true;
        }
        epoch[slot] = cur;
        val[slot] = v;
        return true;
    }
    bool fail(const char *what, u64 pc) {
        std::lock_guard<std::mutex> lk(err_mutex);      // level-parallel runs: several threads may fail at once, the first message stays
        if (!failed) snprintf(err, sizeof(err), "%s at instruction %llu", what, (unsigned long long)pc);
        failed = true;
        return false;
    }

    // the accessors wit_exec works through
    struct IO {
        Runner *r;
        const u32 *is, *os;
        u32 no, k;
        u64 pc;
        u64 in(u32 i) const { return r->val[is[i]]; }
        bool out(u64 v) {
            if (k >= no) {
                k++;
                return true;       // reported as an output count mismatch after the instruction
            }
            return r->set(os[k++], v % GLP, pc);
        }
        // out-of-order outputs (PoseidonGate rows): by index; the instruction then counts as complete
        bool out_at(u32 idx, u64 v) {
            k = no;
            return idx < no ? r->set(os[idx], v % GLP, pc) : r->fail("output index out of range", pc);
        }
        bool fail(int code) { return r->fail(wit_strerror(code), pc); }
    };

    // code (u32 words): [opcode, n_params, n_in, n_out, ins..., outs...] repeated; the parameters of all instructions are
    // consecutive in `params` (64-bit: constants are field elements)
    bool run(const u32 *code, u64 code_len, const int64_t *params, const u32 *in_slots, const u64 *in_vals, u32 n_inputs) {
        cur++;
        failed = false;
        for (u32 i = 0; i < n_inputs; i++)
            if (!set(in_slots[i], in_vals[i] % GLP, 0)) return false;
        u64 pc = 0, ip = 0, pp = 0;
        while (ip < code_len) {
            int op = (int)code[ip];
            u32 np = code[ip + 1], ni = code[ip + 2], no = code[ip + 3];
            const int64_t *pr = params + pp;
            const u32 *is = code + ip + 4, *os = is + ni;
            ip += 4 + ni + no;
            pp += np;
            pc++;
            for (u32 i = 0; i < ni; i++)
                if (epoch[is[i]] != cur) return fail("input not available", pc);
            IO io = {this, is, os, no, 0, pc};
            if (!wit_exec<true>(op, pr, np, ni, no, io)) return false;
            if (io.k != no) return fail("output count mismatch", pc);
        }
        return true;
    }
};

// ---- ONE witness on several host threads: the instructions levelled by data dependence
// The fold of signatures.rs:97-105 is a serial chain: recursion witness -> proof -> next witness.  A recursion circuit's program
// is ~12 k coarse instructions (PoseidonGate rows, reducing / interpolation rows, arithmetic) in ~150 dependence levels, most of the
// time in the ~6 000 Poseidon rows of the 56 Merkle-path chains -- independent of each other.  The plan below assigns every
// instruction the level 1 + max(level of the first writers of its inputs); an instruction that writes a slot somebody wrote before
// (a copy constraint between two generators: the second write only compares) is placed after that first writer.  A level's
// instructions touch disjoint output slots, so the threads of a level share nothing but read-only inputs; a barrier separates levels.
struct LevelPlan {
    std::vector<u64> ip, pp;          // per instruction: offset of its header in `code`, of its parameters in `params`
    std::vector<u32> order;           // instruction indices sorted by level
    std::vector<u32> level_start;     // order[level_start[l] .. level_start[l + 1]) = level l
};
// Plans are cached by the CONTENT of the program (a 64-bit FNV-1a of its words and of its input slots, + lengths): a key on the
// address would hand a stale plan to a different program that numpy later allocates at the same address.  The cache is small and
// bounded (a process holds a few dozen recursion shapes); at the bound it is emptied -- plans in use are kept alive by shared_ptr.
static std::mutex g_plan_mutex;
struct PlanKey {
    u64 hash, code_len;
    u32 n_slots, n_inputs;
    bool operator<(const PlanKey &o) const {
        return std::tie(hash, code_len, n_slots, n_inputs) < std::tie(o.hash, o.code_len, o.n_slots, o.n_inputs);
    }
};
static std::map<PlanKey, std::shared_ptr<LevelPlan>> g_plans;
#define ZKLC_MAX_LEVEL_PLANS 256

// nullptr: the program reads a slot that neither an earlier instruction writes nor the inputs provide -- the serial runner reports
// that as "input not available"; levelled, the reader could land beside its writer.  The caller then uses the serial runner.
static std::shared_ptr<LevelPlan> level_plan(const u32 *code, u64 code_len, u32 n_slots, const u32 *in_slots, u32 n_inputs) {
    u64 h = 1469598103934665603ULL;
    for (u64 i = 0; i < code_len; i++) h = (h ^ code[i]) * 1099511628211ULL;
    for (u32 i = 0; i < n_inputs; i++) h = (h ^ in_slots[i]) * 1099511628211ULL;
    const PlanKey key = {h, code_len, n_slots, n_inputs};
    std::lock_guard<std::mutex> lk(g_plan_mutex);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    std::shared_ptr<LevelPlan> pl = std::make_shared<LevelPlan>();
    std::vector<u32> slot_level(n_slots, 0), level;
    std::vector<char> written(n_slots, 0);
    for (u32 i = 0; i < n_inputs; i++)
        if (in_slots[i] < n_slots) written[in_slots[i]] = 1;            // available at level 0
    u64 ip = 0, pp = 0;
    u32 max_level = 0;
    while (ip < code_len) {
        u32 np = code[ip + 1], ni = code[ip + 2], no = code[ip + 3];
        const u32 *is = code + ip + 4, *os = is + ni;
        u32 lv = 0;
        for (u32 i = 0; i < ni; i++) {
            if (!written[is[i]]) return nullptr;                          // read before any writer: not a levelled program
            lv = lv > slot_level[is[i]] ? lv : slot_level[is[i]];
        }
        lv += 1;
        for (u32 i = 0; i < no; i++)
            if (written[os[i]] && lv <= slot_level[os[i]]) lv = slot_level[os[i]] + 1;    // a second writer compares: after the first
        for (u32 i = 0; i < no; i++)
            if (!written[os[i]]) {
                written[os[i]] = 1;
                slot_level[os[i]] = lv;
            }
        pl->ip.push_back(ip);
        pl->pp.push_back(pp);
        level.push_back(lv);
        max_level = max_level > lv ? max_level : lv;
        ip += 4 + ni + no;
        pp += np;
    }
    // two instructions of one level that both write a slot for the first time in that level would race: push the later one down
    // (cannot happen for the first writer by construction -- `written` is set in program order -- but a later writer of the SAME
    // level as the first one was moved above; nothing else shares a slot)
    u32 n = (u32)level.size();
    std::vector<u32> count(max_level + 2, 0);
    for (u32 i = 0; i < n; i++) count[level[i] + 1]++;
    for (u32 l = 1; l < count.size(); l++) count[l] += count[l - 1];
    pl->level_start.assign(count.begin(), count.end());
    pl->order.resize(n);
    std::vector<u32> cursor(count.begin(), count.end() - 1);
    for (u32 i = 0; i < n; i++) pl->order[cursor[level[i]]++] = i;
    if (g_plans.size() >= ZKLC_MAX_LEVEL_PLANS) g_plans.clear();
    g_plans[key] = pl;
    return pl;
}

struct SpinBarrier {
    std::atomic<u32> count{0}, gen{0};
    u32 n;
    explicit SpinBarrier(u32 n_) : n(n_) {}
    void wait() {
        u32 g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_release);
        } else {
            u32 spins = 0;
            while (gen.load(std::memory_order_acquire) == g)
                if (++spins > 2000) std::this_thread::yield();
        }
    }
};

// the whole program for one witness on `threads` threads; same results and the same failures as Runner::run
static bool run_levelled(Runner &r, const LevelPlan &pl, const u32 *code, const int64_t *params, const u32 *in_slots,
                         const u64 *in_vals, u32 n_inputs, u32 threads) {
    r.cur++;
    r.failed = false;
    for (u32 i = 0; i < n_inputs; i++)
        if (!r.set(in_slots[i], in_vals[i] % GLP, 0)) return false;
    const u32 n_levels = (u32)pl.level_start.size() - 1;
    SpinBarrier bar(threads);
    std::vector<std::atomic<u32>> next(n_levels);
    for (u32 l = 0; l < n_levels; l++) next[l].store(pl.level_start[l], std::memory_order_relaxed);
    std::atomic<bool> stop(false);
    auto worker = [&]() {
        for (u32 l = 0; l < n_levels; l++) {
            const u32 end = pl.level_start[l + 1];
            if (!stop.load(std::memory_order_relaxed))
                for (;;) {
                    u32 k0 = next[l].fetch_add(4, std::memory_order_relaxed);       // chunks of four: Poseidon rows next to one-liners
                    if (k0 >= end) break;
                    u32 k1 = k0 + 4 < end ? k0 + 4 : end;
                    for (u32 k = k0; k < k1; k++) {
                        u32 idx = pl.order[k];
                        u64 ip = pl.ip[idx];
                        int op = (int)code[ip];
                        u32 np = code[ip + 1], ni = code[ip + 2], no = code[ip + 3];
                        const u32 *is = code + ip + 4, *os = is + ni;
                        u64 pc = (u64)idx + 1;
                        bool ok = true;
                        for (u32 i = 0; i < ni && ok; i++)
                            if (r.epoch[is[i]] != r.cur) ok = r.fail("input not available", pc);
                        if (ok) {
                            Runner::IO io = {&r, is, os, no, 0, pc};
                            ok = wit_exec<true>(op, params + pl.pp[idx], np, ni, no, io);
                            if (ok && io.k != no) ok = r.fail("output count mismatch", pc);
                        }
                        if (!ok) stop.store(true, std::memory_order_relaxed);
                    }
                }
            bar.wait();
        }
    };
    std::vector<std::thread> pool;
    for (u32 t = 1; t < threads; t++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return !r.failed;
}

// pool of interpreter states (value + epoch arrays are hundreds of MB for the Ed25519 circuit: keep them across calls)
static std::mutex g_pool_mutex;
static std::vector<Runner *> g_pool;
static Runner *runner_acquire(u32 n_slots) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i]->val.size() == n_slots) {
                Runner *r = g_pool[i];
                g_pool.erase(g_pool.begin() + i);
                return r;
            }
    }
    Runner *r = new Runner();
    r->val.assign(n_slots, 0);
    r->epoch.assign(n_slots, 0);
    return r;
}
static void runner_release(Runner *r) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (g_pool.size() < 16)
        g_pool.push_back(r);
    else
        delete r;
}
// frees the cached interpreter states (hundreds of MB each for the Ed25519 circuit)
extern "C" void zklc_plonky2_witness_release(void) {
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    for (Runner *r : g_pool) delete r;
    g_pool.clear();
}

// Runs the program for `n_witnesses` partial witnesses (input_values: n_witnesses x n_inputs) on up to `threads` host
// threads.  wire_slot / wire_index: the wire cells of every copy class (n_wire_entries), index = col * n_rows + row;
// wires_out: n_witnesses matrices of num_wires x n_rows u64 -- only the listed cells are written (the caller zero-fills the
// buffer once; the set of written cells is the same for every witness of a circuit).  pi_out: n_witnesses x n_pi.
// status[i] = 0 ok / 1 failed; err_out (optional): n_witnesses x 200 bytes of messages.
extern "C" int32_t zklc_plonky2_witness_run(const uint32_t *code, uint64_t code_len, const int64_t *params, uint32_t n_slots,
                                            const uint32_t *input_slots, uint32_t n_inputs, const uint64_t *input_values,
                                            uint32_t n_witnesses, const uint32_t *wire_slot, const uint32_t *wire_index,
                                            uint64_t n_wire_entries, uint32_t num_wires, uint32_t n_rows, uint64_t *wires_out,
                                            const uint32_t *pi_slots, uint32_t n_pi, uint64_t *pi_out, int32_t *status, char *err_out,
                                            uint32_t threads) {
    if (!code || !status || (n_inputs && (!input_slots || !input_values)) || !wires_out) return -1;
    if (threads == 0) threads = 1;
    // one witness, several threads: the levelled form (the serial fold chain); several witnesses: one thread each
    const u32 level_threads = (n_witnesses == 1 && threads > 1) ? (threads > 16 ? 16 : threads) : 1;
    const std::shared_ptr<LevelPlan> plan = level_threads > 1 ? level_plan(code, code_len, n_slots, input_slots, n_inputs) : nullptr;
    if (threads > n_witnesses) threads = n_witnesses ? n_witnesses : 1;
    std::atomic<u32> next(0);
    auto worker = [&]() {
        Runner *rp = runner_acquire(n_slots);
        Runner &r = *rp;
        for (;;) {
            u32 w = next.fetch_add(1);
            if (w >= n_witnesses) break;
            bool ok = plan ? run_levelled(r, *plan, code, params, input_slots, input_values + (size_t)w * n_inputs, n_inputs, level_threads)
                           : r.run(code, code_len, params, input_slots, input_values + (size_t)w * n_inputs, n_inputs);
            u64 *wires = wires_out + (size_t)w * num_wires * n_rows;
            if (ok) {
                auto scatter = [&](u64 k0, u64 k1) {
                    for (u64 k = k0; k < k1; k++) {
                        u32 s = wire_slot[k];
                        if (r.epoch[s] != r.cur) continue;  // unconstrained cell of a class nobody assigned: stays as it is (zero)
                        wires[wire_index[k]] = r.val[s];
                    }
                };
                if (level_threads > 1) {                    // the single witness of the fold chain: scatter on the same threads
                    std::vector<std::thread> sp;
                    u64 per = (n_wire_entries + level_threads - 1) / level_threads;
                    for (u32 t = 1; t < level_threads; t++)
                        sp.emplace_back(scatter, (u64)t * per < n_wire_entries ? (u64)t * per : n_wire_entries,
                                        (u64)(t + 1) * per < n_wire_entries ? (u64)(t + 1) * per : n_wire_entries);
                    scatter(0, per < n_wire_entries ? per : n_wire_entries);
                    for (auto &t : sp) t.join();
                } else {
                    scatter(0, n_wire_entries);
                }
                for (u32 k = 0; k < n_pi; k++) {
                    if (r.epoch[pi_slots[k]] != r.cur) {
                        ok = false;
                        snprintf(r.err, sizeof(r.err), "public input %u was never assigned", k);
                        break;
                    }
                    pi_out[(size_t)w * n_pi + k] = r.val[pi_slots[k]];
                }
            }
            status[w] = ok ? 0 : 1;
            if (err_out) {
                if (ok)
                    err_out[(size_t)w * 200] = 0;
                else
                    memcpy(err_out + (size_t)w * 200, r.err, 200);
            }
        }
        runner_release(rp);
    };
    std::vector<std::thread> pool;
    for (u32 t = 1; t < threads; t++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return 0;
}
