#!/usr/bin/env python3
"""bench.py -- hot-path benchmark of the MI355X zk-light-client backend.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Headline = BASELINE.json's metric on its configs[2]: **Block_i BFT-finality proofs/sec (100 validators)**.  One STEP = one full
`prove_block_bft` (near_bft_finality/src/prove_bft/bft.rs:38-500) of the NEAR mainnet window shipped with the reference
(tests/golden/block_window_HPi5.json: 100 validators, 73 approvals), end to end from the borsh bytes: batched Ed25519
pre-verification, witness generation on the GPU, 73 proofs of the reference's Ed25519 circuit, their left fold and closing proof,
keys / stakes, seven header-hash chains, bp_hash, the joining recursions and the Poseidon-BN128 wrap.  W untimed blocks (the first
builds and uploads every circuit), then exactly K blocks between barriers, max over ranks; the final proof and its wrap are
verified after the timed region (`final_proof_verified`).  Multi-GPU: every rank proves its own block (weak scaling, no data-path
collective); `--scaling strong` proves ONE block with all ranks (signature shards, tree fold, header proofs on the other ranks).

Prints ONE JSON line on rank 0 (the driver contract of the task statement) with
  roofline      the dominant kernel of a block proof (Poseidon leaf hashing), timed live with HIP events in the `merkle` stage:
                algorithmic bytes against the HBM peak, PMC traffic, and `valu` = SQ_INSTS_VALU x 64 / time against the measured
                integer issue ceiling (the resource that binds it)
  cpu_baseline  the oracle's C + OpenMP prover on the host cores (kind = "port"): MEASURED one complete proof at the fold shape
                (bytes compared with the GPU's) and the wires commitment of the Ed25519 shape on bounded samples, the rest of that
                shape SCALED and labelled; `--cpu-baseline-ed25519` measures the whole Ed25519-shape proof instead (minutes)
  stages        msm (BN254 G1 MSM 2^22, C4), lde and merkle (C3 shapes: 234 x 2^17 -> 2^20), prove (per-circuit proof times and the
                Block_i breakdown), bn254_extras (G2 MSM, Fr NTT, pairing checks, a whole Groth16 prove at 2^22), ed25519_verify
                (configs[1]: 8192 approval sets x 100 validators per launch), `--c5-validators N` (synthetic epoch)
`--no-stages`, `--no-prove`, `--no-bn254-extras`, `--no-cpu-baseline` skip parts; `--host-witness` uses the host interpreter.
"""
import argparse
import json
import os
import sys
import time
import threading
T_PROCESS_START = time.perf_counter()
# before anything can initialise the HIP runtime (torch, the library): one hardware queue per stream of the pipeline (zklc_amd/__init__.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

VALIDATORS = 100          # C2: data/validators_ordered.json has 100 entries
BASE_BLOCKS = 4           # distinct signed blocks generated on the host, tiled on the device
MSG_LEN, MSG_STRIDE = 41, 48
BYTES_PER_SIG = 97        # 32 pk + 64 sig + 1 result (message amortised)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s


def make_base_set():
    """100 synthetic validators (SURVEY 8(d) C2 key derivation) signing BASE_BLOCKS
    different Endorsement messages.  Signing uses the oracle (host, untimed)."""
    from oracle import ed25519_ref as ref
    import hashlib
    keys = [ref.synthetic_seed(1, i) for i in range(VALIDATORS)]
    pks = [ref.keypair(k)[2] for k in keys]
    pk_rows, sig_rows, msg_rows = [], [], []
    for b in range(BASE_BLOCKS):
        prev_hash = hashlib.sha256(b"zklc/bench/block" + bytes([b])).digest()
        msg = ref.generate_signed_message(105971806 + b, 105971807 + b, prev_hash)
        assert len(msg) == MSG_LEN
        for v in range(VALIDATORS):
            pk_rows.append(pks[v])
            sig_rows.append(ref.sign(keys[v], msg))
            msg_rows.append(msg + bytes(MSG_STRIDE - MSG_LEN))
    n = len(pk_rows)
    return (np.frombuffer(b"".join(pk_rows), np.uint8).reshape(n, 32).copy(),
            np.frombuffer(b"".join(sig_rows), np.uint8).reshape(n, 64).copy(),
            np.frombuffer(b"".join(msg_rows), np.uint8).reshape(n, MSG_STRIDE).copy())


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


class GpuTelemetry:
    """Samples shader clock, power, temperature and busy percentage of every amdgpu card in sysfs on a background thread, so that
    the bench line can say WHY a sustained run differs from a short one (DVFS under a power / thermal budget, host stalls):
    `mark()` closes a step and returns the means since the previous mark FOR THE BUSIEST CARD of that interval (a box may expose
    cards this process does not use, and sysfs card order is not the HIP device order).  Absent files give no key."""

    KEYS = (("sclk_mhz", ("freq1_input",), 1e-6), ("power_w", ("power1_average", "power1_input"), 1e-6),
            ("temp_c", ("temp2_input", "temp1_input"), 1e-3), ("mclk_mhz", ("freq2_input",), 1e-6))

    def __init__(self, period=0.2, pci=None):
        """pci: "dddd:bb:dd.f" of the device this process computes on (own_pci()): only that card is sampled.  Round 6 found the
        round-5 line's 2155 MHz / 1.23 kW next to an unchanged block time to be ANOTHER tenant's card: the boxes expose every GPU of
        the node in sysfs and "the busiest card" is not necessarily ours (profiles/r06a_*: a 324 W, 100 % busy neighbour)."""
        import glob
        import threading
        self.cards = []
        self.own = None
        for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            if not os.path.exists(os.path.join(d, "gpu_busy_percent")):
                continue
            if pci is not None:
                if os.path.basename(os.path.realpath(d)).lower() != pci.lower():
                    continue
                self.own = pci
            files = {"busy_pct": (os.path.join(d, "gpu_busy_percent"), 1.0)}
            for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                for key, names, scale in self.KEYS:
                    for nm in names:
                        if key not in files and os.path.exists(os.path.join(h, nm)):
                            files[key] = (os.path.join(h, nm), scale)
            self.cards.append({"name": os.path.basename(os.path.dirname(d)), "files": files, "acc": {}, "n": 0})
        self.period = period
        self.lock, self.stop_ev = threading.Lock(), threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        while not self.stop_ev.wait(self.period):
            for c in self.cards:
                r = {}
                for k, (f, scale) in c["files"].items():
                    try:
                        r[k] = float(open(f).read().split()[0]) * scale
                    except (OSError, ValueError, IndexError):
                        pass
                with self.lock:
                    for k, v in r.items():
                        c["acc"][k] = c["acc"].get(k, 0.0) + v
                    c["n"] += 1

    def mark(self):
        with self.lock:
            best = None
            for c in self.cards:
                m = {k: round(v / max(1, c["n"]), 1) for k, v in c["acc"].items()}
                m["samples"], m["card"] = c["n"], c["name"]
                m["card_selected_by"] = "pci address" if self.own else "busiest card (own PCI address unknown)"
                if best is None or (m.get("busy_pct", 0), m.get("power_w", 0)) > (best.get("busy_pct", 0), best.get("power_w", 0)):
                    best = m
                c["acc"], c["n"] = {}, 0
        return best or {"samples": 0}

    def close(self):
        self.stop_ev.set()

    @staticmethod
    def own_pci(device=None):
        """PCI address of the HIP device this process uses, as sysfs spells it; None when torch cannot say"""
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch.cuda.current_device() if device is None else device)
            return "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            return None

    @staticmethod
    def smi_snapshot():
        """one `rocm-smi` reading (clocks, power, temperature) as text lines; None when the tool is absent"""
        import subprocess
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "-d", "0"], capture_output=True, text=True, timeout=20)
            keep = [ln.strip() for ln in r.stdout.splitlines() if any(w in ln for w in ("sclk", "mclk", "Power", "junction"))]
            return keep[:8] or None
        except (OSError, subprocess.SubprocessError):
            return None


def rss_mb():
    try:
        import psutil
        return round(psutil.Process().memory_info().rss / 2**20, 1)
    except Exception:
        return None


def pmc_traffic(n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (bench.py cannot read
    hardware counters itself); None when the profile was taken at another batch size."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "ed25519_pmc_latest.json")))
        return j["hbm_bytes_per_launch"] if j["signatures"] == n else None
    except (OSError, ValueError, KeyError):
        return None


# Issue limits of the SIMDs (round 5: a hardware denominator instead of the empirical one of rounds 2-4).  A wave64 instruction of the
# multi-pass integer class (v_mad_u64_u32, v_mul_*, carry adds, v_cndmask with an SGPR mask, 64-bit shifts) occupies a SIMD for 4
# cycles, one of the single-pass class (v_mov / v_add_u32 / v_xor / 32-bit shifts) for 2: 1024 SIMDs x 64 lanes x 2.4 GHz / 4 = 39.3 T
# lane-instructions/s, / 2 = 78.6 T.  The micro-benchmark (tools/ubench/valu_ubench.hip, warm clock, sclk sampled:
# profiles/r05a_valu_ubench_warm_sclk.txt) reaches 4.3-4.4 and 2.4-2.6 cycles: 36.0-36.8 T and 60-66 T -- carried as `ubench`.
VALU_CLOCK_GHZ = 2.4
VALU_INT_PEAK_TLOPS = 1024 * 64 * VALU_CLOCK_GHZ / 4 / 1e3
VALU_FAST_PEAK_TLOPS = 1024 * 64 * VALU_CLOCK_GHZ / 2 / 1e3
VALU_INT_UBENCH_TLOPS, VALU_FAST_UBENCH_TLOPS = 36.3, 61.0
# static share of single-pass instructions in each priced kernel (tools/valu_mix.py over the shipped code objects, profiles/r04_valu_mix.txt)
VALU_FAST_SHARE = {"poseidon": 0.001, "lde": 0.136, "msm": 0.202}


def poseidon_pmc():
    """counters of gl_hash_leaves_kernel at the bench's Merkle shape from the committed rocprofv3 PMC passes
    (tools/pmc_merkle.sh -> profiles/poseidon_pmc_latest.json); None when absent"""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "poseidon_pmc_latest.json")))
        return j if j.get("leaves") == 1 << 20 and j.get("width") == 234 else None
    except (OSError, ValueError):
        return None


def pmc_json(name):
    """a committed rocprofv3 PMC summary (profiles/<name>, written by tools/pmc_*.sh on the GPU box); None when absent"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None


def valu_block(wave_instructions, ms, what, kernel="poseidon"):
    """the binding resource of the integer kernels: SQ_INSTS_VALU x 64 lanes / time against the issue ceiling of the kernel's
    instruction mix, 1 / ((1 - f) / int-class rate + f / single-pass rate) with f the static single-pass share"""
    ach = wave_instructions * 64 / (ms * 1e-3) / 1e12
    f = VALU_FAST_SHARE[kernel]
    peak = 1.0 / ((1.0 - f) / VALU_INT_PEAK_TLOPS + f / VALU_FAST_PEAK_TLOPS)
    ub = 1.0 / ((1.0 - f) / VALU_INT_UBENCH_TLOPS + f / VALU_FAST_UBENCH_TLOPS)
    return {"bound": "valu-int", "achieved": ach, "peak": peak, "unit": "T lane-instr/s", "frac": ach / peak, "ubench": ub,
            "note": "SQ_INSTS_VALU (%s) x 64 lanes / the live time; peak = the SIMDs' issue limit at 2.4 GHz (4 cycles per multi-pass "
                    "wave64 instruction, 2 per single-pass one) weighted by the kernel's static instruction mix (single-pass share "
                    "%.3f, tools/valu_mix.py); ubench = what the micro-benchmark reaches with the same mix "
                    "(profiles/r05a_valu_ubench_warm_sclk.txt)" % (what, f)}


def cpu_baseline(pk, sg, ms, budget_s=12.0):
    """Oracle C restatement (oracle/c/ed25519_oracle.c) on all host cores, bounded sample."""
    from oracle import cport
    threads = host_cores()
    reps = 20
    pkb, sgb, msb = np.tile(pk, (reps, 1)), np.tile(sg, (reps, 1)), np.tile(ms, (reps, 1))
    n = pkb.shape[0]
    cport.ed25519_verify_batch(pkb[:256], sgb[:256], msb[:256], MSG_LEN, MSG_STRIDE, 256, nthreads=threads)  # warm
    done, t0 = 0, time.perf_counter()
    while True:
        ok, used = cport.ed25519_verify_batch(pkb, sgb, msb, MSG_LEN, MSG_STRIDE, n, nthreads=threads)
        assert int(ok.sum()) == n
        done += n
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            break
    return {"value": done / dt, "unit": "sig/s", "cores": used, "kind": "port",
            "sample": "%d signatures (the bench's %d-signature base set repeated), oracle/c/ed25519_oracle.c, "
                      "gcc -O3 -fopenmp, %.1f s" % (done, pk.shape[0], dt)}


def _time_stream(fn, stream, iters, barrier):
    """average ms of `fn` over `iters` launches on `stream` (HIP events), after one warm-up"""
    import torch
    fn()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    barrier()
    wall = (time.perf_counter() - t0) / iters * 1e3
    return e0.elapsed_time(e1) / iters, wall


def run_stages(args, ctx, dev, stream, rank, world, with_cpu):
    """Secondary measurements: MSM (C4), LDE and Merkle commit (C3).  Returns a dict on every rank."""
    import torch
    import torch.distributed as dist
    from oracle import cport
    import zklc_amd

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x):
        if world > 1:
            t = torch.tensor([x], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t[0])
        return x

    res = {}
    threads = host_cores()
    # ---------------- MSM: 2^msm_log points per GPU
    n = 1 << args.msm_log
    pts_h = cport.bn254_gen_points(n, 5 + 1000003 * rank, 3)      # (5 + 1000003 rank + 3 i) * G
    rng = np.random.default_rng(1 + rank)
    sc_h = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    sc_h[:, 3] &= np.uint64((1 << 60) - 1)                        # < 2^252 < r
    d_pts = torch.from_numpy(pts_h.view(np.int64)).to(dev)
    d_sc = torch.from_numpy(sc_h.view(np.int64)).to(dev)
    wb = ctx.bn254_g1_msm_workspace_bytes(n)
    d_ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    d_out = torch.zeros(9, dtype=torch.int64, device=dev)         # 8 words + infinity flag (as int64 for the gather)
    d_inf = torch.zeros(2, dtype=torch.int32, device=dev)
    gathered = [torch.zeros(9, dtype=torch.int64, device=dev) for _ in range(world)]
    d_cpts = torch.zeros((world, 8), dtype=torch.int64, device=dev)
    d_ones = torch.zeros((world, 4), dtype=torch.int64, device=dev)
    d_ones[:, 0] = 1
    wb2 = ctx.bn254_g1_msm_workspace_bytes(world)
    d_ws2 = torch.empty(wb2, dtype=torch.uint8, device=dev)
    d_final = torch.zeros(8, dtype=torch.int64, device=dev)

    def msm_step(d_pts=d_pts, d_sc=d_sc, n=n):
        ctx.bn254_g1_msm_dev(d_pts, d_sc, n, d_out, d_inf, d_ws, wb, stream=stream)
        if world > 1:
            with torch.cuda.stream(stream):
                d_out[8] = d_inf[0]
                if dist.get_backend() == "nccl":
                    dist.all_gather(gathered, d_out)
                    g = torch.stack(gathered)
                else:       # gloo (functional runs with several ranks on one GPU): the 72-byte partials travel through the host
                    stream.synchronize()
                    hs = [torch.zeros(9, dtype=torch.int64) for _ in range(world)]
                    dist.all_gather(hs, d_out.cpu())
                    g = torch.stack(hs).to(dev)
                d_cpts.copy_(g[:, :8] * (g[:, 8:9] == 0))            # infinity partials -> (0, 0)
            ctx.bn254_g1_msm_dev(d_cpts, d_ones, world, d_final, d_inf[1:], d_ws2, wb2, stream=stream)

    ms, wall = _time_stream(msm_step, stream, 3, barrier)
    ms, wall = reduce_max(ms), reduce_max(wall)
    msm = {"metric": "BN254 G1 MSM", "value": n * world / (wall * 1e-3) / 1e6, "unit": "Melem/s", "points_per_gpu": n,
           "ms": wall, "kernel_ms": ms,
           "roofline": {"bound": "hbm", "achieved": 96.0 * n / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": 96.0 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "note": "algorithmic bytes = 96 B/element (64 B point + 32 B scalar); integer-VALU-bound"}}
    # the other scalar distributions of SURVEY 8(d) C4 at the same size (a Groth16 witness is W-shaped: gnark-plonky2-verifier/cmd/
    # web-api.go:77): (W) 50 % in {0, 1}, 30 % < 2^64, 20 % uniform; (A1) all scalars equal; (A2) all < 2^64.  Parity of these shapes
    # against the oracle is tests/test_gpu_bn254.py (2^20); here the result must not depend on the order of the points
    by_dist = {}
    for name in ("W", "A1", "A2"):
        s_ = sc_h.copy()
        if name == "W":
            kind = rng.random(n)
            small = kind < 0.5
            s_[small] = 0
            s_[small, 0] = rng.integers(0, 2, size=int(small.sum()), dtype=np.uint64)
            s_[(kind >= 0.5) & (kind < 0.8), 1:] = 0
        elif name == "A1":
            s_[:] = s_[0]
        else:
            s_[:, 1:] = 0
        d_s2 = torch.from_numpy(s_.view(np.int64)).to(dev)
        ms_d, wall_d = _time_stream(lambda: msm_step(d_pts, d_s2, n), stream, 3, barrier)
        wall_d = reduce_max(wall_d)
        by_dist[name] = {"ms": wall_d, "value": n * world / (wall_d * 1e-3) / 1e6, "unit": "Melem/s"}
        del d_s2
    msm["distributions"] = by_dist
    msm_step()                                                    # d_out holds the uniform instance again (parity check below)
    # the FIXED-BASE form (round 5, include/zklc.h): the bases of a Groth16 proving key are the same for every proof, so a table of
    # 2^(c w) P_i is built once and a multi-exponentiation needs no closing doublings and one bucket reduction; its affine result is
    # bit-identical to the plain form's.  Reported beside the plain figure (which stays the headline's MSM number).
    if world == 1 and args.extra_stages:
        try:
            torch.cuda.synchronize()
            plain_words = d_out[:8].cpu().numpy().copy()
            t_ = time.perf_counter()
            table = ctx.bn254_msm_fixed_table(d_pts, n, stream=stream)
            stream.synchronize()
            t_tab = time.perf_counter() - t_
            fstep = lambda sc_=d_sc: ctx.bn254_msm_fixed_dev(table, sc_, n, d_out, d_inf, d_ws, wb, stream=stream)
            ms_f, wall_f = _time_stream(fstep, stream, 3, barrier)
            stream.synchronize()
            same = bool(np.array_equal(d_out[:8].cpu().numpy(), plain_words))
            msm["fixed_base"] = {"metric": "BN254 G1 MSM over a precomputed table of the bases (2^(16 w) P_i, one bucket set)", "ms": wall_f,
                                 "value": n / (wall_f * 1e-3) / 1e6, "unit": "Melem/s", "table_mb": table.numel() / 1e6,
                                 "table_build_s": t_tab, "equals_plain_result": same}
            s_w = sc_h.copy()
            kind = np.random.default_rng(5).random(n)
            s_w[kind < 0.5] = 0
            s_w[kind < 0.5, 0] = 1
            s_w[(kind >= 0.5) & (kind < 0.8), 1:] = 0
            d_sw = torch.from_numpy(s_w.view(np.int64)).to(dev)
            _, wall_fw = _time_stream(lambda: fstep(d_sw), stream, 3, barrier)
            msm["fixed_base"]["witness_like_ms"] = wall_fw
            del table, d_sw
            msm_step()
        except Exception as e:      # an extra line of the report
            msm["fixed_base"] = {"error": repr(e)[:200]}
    pm = pmc_json("msm_pmc_latest.json")
    if pm is not None and pm.get("log_n") == args.msm_log:
        msm["roofline"]["traffic"] = pm.get("hbm_bytes_per_msm")
        msm["roofline"]["valu"] = valu_block(pm["valu_wave_instructions_per_msm"], ms, "all kernels of one multi-exponentiation, "
                                                                                       "profiles/msm_pmc_latest.json", "msm")
    if with_cpu:
        t0 = time.perf_counter()
        want, winf, used = cport.bn254_msm(pts_h, sc_h, nthreads=threads)
        dt = time.perf_counter() - t0
        got = d_out[:8].cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want), "GPU MSM differs from the oracle"
        msm["cpu_baseline"] = {"value": n / dt / 1e6, "unit": "Melem/s", "cores": used, "kind": "port",
                               "sample": "the same 2^%d-point MSM once, oracle/c/bn254_oracle.c (also the parity check)" % args.msm_log}
    if world > 1:
        # STRONG form (distributed.msm_sharded): ONE 2^msm_log instance, index-sharded over the ranks; every rank reduces its
        # shard to one point, the partials are all-gathered (world x 72 bytes over RCCL) and added locally.  Rank 0 also computes the
        # whole instance alone: the sharded result must be the same affine point.
        import importlib
        DIST = importlib.import_module("zk-light-client-implementation_amd.distributed")
        pts_all = cport.bn254_gen_points(n, 5, 3)
        rng1 = np.random.default_rng(1)
        sc_all = rng1.integers(0, 2**63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng1.integers(0, 2, size=(n, 4), dtype=np.uint64)
        sc_all[:, 3] &= np.uint64((1 << 60) - 1)
        lo_, hi_ = DIST.shard_range(n, rank, world)
        d_ps = torch.from_numpy(pts_all[lo_:hi_].view(np.int64).copy()).to(dev)
        d_ss = torch.from_numpy(sc_all[lo_:hi_].view(np.int64).copy()).to(dev)
        step_s = lambda: msm_step(d_ps, d_ss, hi_ - lo_)
        ms_s, wall_s = _time_stream(step_s, stream, 3, barrier)
        ms_s, wall_s = reduce_max(ms_s), reduce_max(wall_s)
        sharded = d_final.cpu().numpy().view(np.uint64).copy()
        same = None
        if rank == 0:
            d_pa, d_sa = torch.from_numpy(pts_all.view(np.int64)).to(dev), torch.from_numpy(sc_all.view(np.int64)).to(dev)
            ctx.bn254_g1_msm_dev(d_pa, d_sa, n, d_out, d_inf, d_ws, wb, stream=stream)
            torch.cuda.synchronize()
            same = bool(np.array_equal(d_out[:8].cpu().numpy().view(np.uint64), sharded))
            del d_pa, d_sa
        msm["strong"] = {"metric": "BN254 G1 MSM, ONE 2^%d instance index-sharded over %d GPUs (all-gather of the partial sums)" % (args.msm_log, world),
                         "value": n / (wall_s * 1e-3) / 1e6, "unit": "Melem/s", "ms": wall_s, "kernel_ms": ms_s,
                         "equals_single_gpu_result": same}
        del d_ps, d_ss, pts_all, sc_all
    res["msm"] = msm
    del d_pts, d_sc, d_ws, pts_h, sc_h
    torch.cuda.empty_cache()

    # ---------------- LDE + Merkle commit at the C3 shape
    log_n, rate, batch, cap = 17, 3, 234, 4
    nn, N = 1 << log_n, 1 << (log_n + rate)
    g = torch.Generator(device=dev).manual_seed(0xC0FFEE + rank)
    coeffs = torch.randint(0, 2**63 - 1, (batch, nn), generator=g, device=dev, dtype=torch.int64)  # < p: canonical
    lde = torch.empty((batch, N), dtype=torch.int64, device=dev)
    ms, wall = _time_stream(lambda: ctx.gl_lde_dev(coeffs, log_n, rate, batch, 7, lde, flags=zklc_amd._lib.NTT_OUT_BITREV, stream=stream),
                            stream, 5, barrier)
    ms = reduce_max(ms)
    alg = 8.0 * (nn + N) * batch
    res["lde"] = {"metric": "Goldilocks coset LDE 234 x (2^17 -> 2^20)", "value": world * alg / (ms * 1e-3) / 1e9, "unit": "GB/s",
                  "ms": ms, "gbutterflies_per_s": world * (N // 2) * (log_n + rate) * batch / (ms * 1e-3) / 1e9,
                  "roofline": {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                               "note": "algorithmic bytes = 8*(n + N) per polynomial (read coefficients once, write evaluations once)"}}
    pl_ = pmc_json("lde_pmc_latest.json")
    if pl_ is not None:
        res["lde"]["roofline"]["traffic"] = pl_.get("hbm_bytes_per_lde")
        res["lde"]["roofline"]["valu"] = valu_block(pl_["valu_wave_instructions_per_lde"], ms, "all passes of one extension, "
                                                                                                "profiles/lde_pmc_latest.json", "lde")
    # the PRODUCT shape: every commitment of the Ed25519 circuit extends 2^18 -> 2^21 (prove_crypto/ed25519.rs:60), the less efficient
    # case of the two: PMC at this shape in profiles/lde_pmc_2p18_latest.json
    try:
        c18 = torch.randint(0, 2**63 - 1, (batch, 1 << 18), generator=g, device=dev, dtype=torch.int64)
        l21 = torch.empty((batch, 1 << 21), dtype=torch.int64, device=dev)
        ms18, _ = _time_stream(lambda: ctx.gl_lde_dev(c18, 18, rate, batch, 7, l21, flags=zklc_amd._lib.NTT_OUT_BITREV, stream=stream),
                               stream, 5, barrier)
        ms18 = reduce_max(ms18)
        alg18 = 8.0 * ((1 << 18) + (1 << 21)) * batch
        res["lde_2p18"] = {"metric": "Goldilocks coset LDE 234 x (2^18 -> 2^21), the Ed25519 circuit's shape", "value": world * alg18 / (ms18 * 1e-3) / 1e9,
                           "unit": "GB/s", "ms": ms18,
                           "roofline": {"bound": "hbm", "achieved": alg18 / (ms18 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": alg18 / (ms18 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None}}
        p18 = pmc_json("lde_pmc_2p18_latest.json")
        if p18 is not None:
            res["lde_2p18"]["roofline"]["traffic"] = p18.get("hbm_bytes_per_lde")
            res["lde_2p18"]["roofline"]["valu"] = valu_block(p18["valu_wave_instructions_per_lde"], ms18, "all passes of one extension, "
                                                             "profiles/lde_pmc_2p18_latest.json", "lde")
        del c18, l21
    except KeyError:
        pass
    except Exception as e:          # an extra line of the report, never a reason to lose the bench
        res["lde_2p18"] = {"error": repr(e)[:200]}
    words = ctx.gl_merkle_tree_words(log_n + rate, cap)
    tree = torch.empty(words, dtype=torch.int64, device=dev)
    ms, wall = _time_stream(lambda: ctx.gl_merkle_commit_dev(lde, N, log_n + rate, batch, cap, tree, stream=stream), stream, 3, barrier)
    ms = reduce_max(ms)
    algm = (8.0 * batch + 32) * N
    res["merkle"] = {"metric": "Poseidon Merkle commit, 2^20 leaves x 234 columns, cap 4", "value": world * N / (ms * 1e-3) / 1e6,
                     "unit": "Mleaf/s", "ms": ms, "mperm_per_s": world * (N * ((batch + 7) // 8) + N) / (ms * 1e-3) / 1e6,
                     "roofline": {"bound": "hbm", "achieved": algm / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": algm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                  "note": "algorithmic bytes = 8*width + 32 per leaf; 30 Poseidon permutations per leaf: VALU-bound"}}
    if with_cpu:
        cb, cl = 16, 16
        ch = coeffs[:cb].cpu().numpy().view(np.uint64)
        t0 = time.perf_counter()
        ref_lde = cport.gl_lde(ch, rate, 7, nthreads=threads)
        dt = time.perf_counter() - t0
        br = np.array([int(format(i, "020b")[::-1], 2) for i in range(N)])
        assert np.array_equal(lde[:cb].cpu().numpy().view(np.uint64), ref_lde[:, br]), "GPU LDE differs from the oracle"
        res["lde"]["cpu_baseline"] = {"value": 8.0 * (nn + N) * cb / dt / 1e9, "unit": "GB/s", "cores": threads, "kind": "port",
                                      "sample": "%d of the 234 polynomials, oracle/c/goldilocks_oracle.c (also the parity check)" % cb}
        sub = lde[:, :1 << cl].cpu().numpy().view(np.uint64).copy()
        t0 = time.perf_counter()
        lv = cport.gl_merkle_commit(sub, cap, nthreads=threads)
        dt = time.perf_counter() - t0
        assert np.array_equal(tree[:4 << cl].cpu().numpy().view(np.uint64).reshape(-1, 4), lv[0]), "GPU leaf digests differ from the oracle"
        res["merkle"]["cpu_baseline"] = {"value": (1 << cl) / dt / 1e6, "unit": "Mleaf/s", "cores": threads, "kind": "port",
                                         "sample": "the first 2^%d leaves (x 234 columns), oracle/c/goldilocks_oracle.c" % cl}
    if not args.no_prove:
        del coeffs, lde, tree
        torch.cuda.empty_cache()
        res["prove"] = run_prove_stage(args, ctx, dev, stream, rank, world, barrier, reduce_max)
        if with_cpu and "cpu_baseline" in res["lde"] and "cpu_baseline" in res["merkle"]:
            # CPU lower bound of ONE Ed25519 proof (2^18 rows) with the oracle's C port: only its wires commitment
            # (LDE of 234 columns 2^18 -> 2^21 + Merkle tree over 2^21 leaves), extrapolated from the two bounded samples above
            lde_s = 2 * alg / 1e9 / res["lde"]["cpu_baseline"]["value"]
            mk_s = 2 * N / 1e6 / res["merkle"]["cpu_baseline"]["value"]
            try:
                gpu_ms = res["prove"]["ed25519_circuit_2p18x234"]["ms_per_proof"]
                fold = next((v["cpu_baseline"] for k, v in res["prove"].items()
                             if k.startswith("recursion_fold_R(R,ed)") and isinstance(v, dict) and "cpu_baseline" in v), None)
                ed_cb = {"cores": threads, "kind": "port",
                         "measured": "the wires commitment of this shape: coset LDE %.1f s + Poseidon Merkle tree %.1f s, from the bounded lde / "
                                     "merkle samples above (oracle/c/goldilocks_oracle.c)" % (lde_s, mk_s)}
                meas = res["prove"]["ed25519_circuit_2p18x234"].get("cpu_baseline_measured")
                if meas is not None:       # --cpu-baseline-ed25519: nothing scaled
                    ed_cb.update(meas)
                    ed_cb["measured"] = meas["sample"]
                elif fold is not None:
                    st = fold["stages_s"]
                    ratio = st["proof"] / st["wires_commit"]
                    ed_cb.update({"value": 1.0 / ((lde_s + mk_s) * ratio), "unit": "proofs/s (wires commitment measured, rest scaled)",
                                  "scaled": "the other stages (Z / partial products, quotient, openings, FRI) by the ratio whole proof / wires "
                                            "commitment = %.2f of the COMPLETE C proof measured at the fold shape" % ratio,
                                  "sample": "wires commitment measured on bounded samples, other stages scaled from the complete CPU proof of the "
                                            "fold shape", "seconds_per_proof": (lde_s + mk_s) * ratio})
                else:
                    ed_cb.update({"value": 1.0 / (lde_s + mk_s), "unit": "proofs/s (upper bound)",
                                  "sample": "wires commitment only: a LOWER bound of the time of one CPU proof with this port",
                                  "seconds_per_proof": lde_s + mk_s})
                ed_cb["gpu_speedup"] = ed_cb["seconds_per_proof"] * 1e3 / gpu_ms
                res["prove"]["ed25519_circuit_2p18x234"]["cpu_baseline"] = ed_cb
            except Exception as e:      # a report, never a reason to lose the bench line
                res["prove"]["ed25519_circuit_2p18x234"]["cpu_baseline_error"] = repr(e)[:300]
    if not args.no_bn254_extras:
        try:
            res["bn254_extras"] = run_bn254_extras(ctx, dev, reduce_max, barrier, world)
        except Exception as e:          # secondary figures: never a reason to lose the line
            if world > 1:
                raise                   # the other ranks are in a collective: fail the job rather than leave them waiting
            res["bn254_extras_error"] = repr(e)[:300]
    return res


def pts1_256(B):
    """256 distinct G1 points in gnark's layout (an arithmetic progression on the curve; host, untimed)"""
    cur, step, out = B.mul(54321, B.G1), B.mul(991, B.G1), []
    for _ in range(256):
        out.append(B.to_mont_words(cur[0]) + B.to_mont_words(cur[1]))
        cur = B.add(cur, step)
    return out


def run_bn254_extras(ctx, dev, reduce_max, barrier, world):
    """G2 MSM, Fr coset NTT and Groth16-shaped pairing checks (the rest of the Groth16 wrap)"""
    import torch
    import zklc_amd
    from oracle import bn254 as B
    lib = zklc_amd.load()
    sp = ctx.stream_ptr()
    out = {}

    def timed(fn, reps=3):
        fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return reduce_max((time.perf_counter() - t0) / reps * 1e3)
    rng = np.random.default_rng(3)
    # G2 MSM 2^18 (256 distinct points tiled: generating G2 points on the host is slow)
    cur, step, pts = B.g2_mul(12345, B.G2), B.g2_mul(777, B.G2), []
    for _ in range(256):
        pts.append(B.g2_to_words(cur))
        cur = B.g2_add(cur, step)
    n = 1 << 18
    d_p = torch.from_numpy(np.tile(np.array(pts, dtype=np.uint64), (n // 256, 1)).view(np.int64)).to(dev)
    sc = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 60) - 1)
    d_s = torch.from_numpy(sc.view(np.int64)).to(dev)
    wb = int(lib.zklc_bn254_g2_msm_workspace_bytes(n))
    d_w = torch.empty(wb, dtype=torch.uint8, device=dev)
    d_o = torch.zeros(17, dtype=torch.int64, device=dev)
    ms = timed(lambda: ctx._check(lib.zklc_bn254_g2_msm_dev(ctx._h, sp, d_p.data_ptr(), d_s.data_ptr(), n, d_o.data_ptr(),
                                                            d_o.data_ptr() + 128, d_w.data_ptr(), wb)))
    out["g2_msm_2p18"] = {"ms": ms, "value": world * n / ms / 1e3, "unit": "Melem/s"}
    del d_p, d_s, d_w
    # Fr coset NTT 2^22
    n = 1 << 22
    a = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    d_a = torch.from_numpy(a.view(np.int64)).to(dev)
    wb = int(lib.zklc_bn254_fr_ntt_workspace_bytes(22))
    d_w = torch.empty(wb, dtype=torch.uint8, device=dev)
    ms = timed(lambda: ctx._check(lib.zklc_bn254_fr_ntt_dev(ctx._h, sp, d_a.data_ptr(), 22, 0, 1, d_w.data_ptr(), wb)))
    out["fr_coset_ntt_2p22"] = {"ms": ms, "value": world * 64.0 * n / ms / 1e6, "unit": "GB/s (64 B/element algorithmic)"}
    del d_a, d_w
    # pairing checks, k = 4 (Groth16 verification shape)
    p, q = B.mul(0xABCDEF, B.G1), B.g2_mul(0x13579B, B.G2)
    one = [(p, q), (B.neg(p), q), (B.mul(5, B.G1), B.G2), (B.neg(B.G1), B.g2_mul(5, B.G2))]
    checks = 4096
    g1 = np.array([[B.to_mont_words(x[0]) + B.to_mont_words(x[1]) for x, _ in one]] * checks, dtype=np.uint64)
    g2 = np.array([[B.g2_to_words(y) for _, y in one]] * checks, dtype=np.uint64)
    d1, d2 = torch.from_numpy(g1.view(np.int64)).to(dev), torch.from_numpy(g2.view(np.int64)).to(dev)
    d_r = torch.zeros(checks, dtype=torch.int32, device=dev)
    ms = timed(lambda: ctx._check(lib.zklc_bn254_pairing_check_dev(ctx._h, sp, d1.data_ptr(), d2.data_ptr(), 4, checks, d_r.data_ptr(), None)), reps=2)
    assert int(d_r.sum()) == checks
    out["pairing_checks_k4_x4096"] = {"ms": ms, "value": world * checks / ms * 1e3, "unit": "checks/s"}
    del d1, d2, d_r
    # the whole of `groth16.Prove` at the size BASELINE configs[3] names (2^22 constraints / wires): computeH (seven Fr NTTs + the
    # pointwise quotient in HBM) and the four multi-exponentiations, through zklc_amd.groth16.Groth16Prover (row a10).  Synthetic
    # key (256 distinct points tiled) and random operands: the proof is not meaningful, the work is the real prover's.
    from zklc_amd.groth16 import Groth16Prover, fr_to_mont_words
    lg = 22
    n = 1 << lg
    g1p = np.array(pts1_256(B), dtype=np.uint64)
    g2p = np.array(pts, dtype=np.uint64)
    tile1 = lambda k: np.tile(g1p, ((k + 255) // 256, 1))[:k]
    pk = {"n": n, "n_public": 4, "A_words": tile1(n), "B1_words": tile1(n), "K_words": tile1(n - 5), "Z_words": tile1(n - 1),
          "B2_words": np.tile(g2p, (n // 256, 1)), "alpha1_words": g1p[1:2], "beta1_words": g1p[2:3], "delta1_words": g1p[3:4],
          "beta2_words": g2p[1:2], "delta2_words": g2p[2:3]}
    t0 = time.perf_counter()
    gp = Groth16Prover(ctx, pk)
    t_up = time.perf_counter() - t0
    del pk
    w = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64)
    w[:, 3] &= np.uint64((1 << 60) - 1)
    w[0] = [1, 0, 0, 0]
    abc = tuple(rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) & np.uint64((1 << 61) - 1) for _ in range(3))
    gp.prove_words(w, abc, 12345, 67890)
    t0 = time.perf_counter()
    gp.prove_words(w, abc, 12345, 67890)
    dt = time.perf_counter() - t0
    out["groth16_prove_2p22"] = {"ms": reduce_max(dt * 1e3), "value": world / dt, "unit": "proofs/s", "stages_ms": dict(gp.last_ms),
                                 "key_upload_s_untimed": t_up,
                                 "published_cpu": "30 s per Groth16 proof on a 16-core Ryzen 9 7950X (gnark-plonky2-verifier/README.md:35-39; "
                                                  "the only number the reference publishes for this step; its circuit size is not stated)",
                                 "note": "2^22 wires / constraints, operands handed over as host arrays (PCIe inside), synthetic key"}
    gp.close()
    return out


def run_prove_stage(args, ctx, dev, stream, rank, world, barrier, reduce_max):
    """plonky2 proofs of the reference's circuits through the package's pipeline (zklc_amd.pipeline.BlockPipeline): per-circuit proof
    times (the per-signature Ed25519 circuit, the recursion shapes of the fold, the closing proof and the BN128 wrap, SHA-256, keys /
    stakes), then the timed Block_i proofs."""
    import hashlib
    import torch
    import zklc_amd
    from zklc_amd.pipeline import BlockPipeline, BlockWindow
    from zklc_amd.plonky2 import HASH_GL, HASH_BN128, ed25519_circuit as E
    from zklc_amd.plonky2 import serialization as S
    out = {}
    ed_name = "ed25519_circuit_2p18x234"
    t_setup = time.perf_counter()
    pipe = BlockPipeline(torch.cuda.current_device(), prove_streams=args.prove_streams, witness_batch=args.witness_batch,
                         host_witness=args.host_witness, rank=rank, world=world, comm_device=dev, host_threads=host_cores(),
                         device_share=-(-world // max(1, torch.cuda.device_count())))

    # cold start: the cacheable circuits the run needs (Ed25519, the SHA-256 circuits of the header chains) that are not in the
    # circuit cache yet are built by worker processes side by side, not one after the other under this process's GIL
    win_ = BlockWindow.from_fixture(json.load(open(os.path.join(ROOT, "tests", "golden", "block_window_HPi5.json"))))
    c1_msg_len = len(bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "ed25519_near_c1_small.json")))["msg"]))
    out["circuit_prewarm"] = pipe.prewarm(win_, extra_msg_lens=[c1_msg_len], timeout_s=300)
    if world > 1:
        barrier()

    def time_proof(prover, wires, pis, sp, bits):
        d_w = torch.from_numpy(np.ascontiguousarray(wires).view(np.int64)).to(dev)
        fn = lambda: prover.prove_dev(d_w.data_ptr(), pis, stream=sp)
        fn()
        barrier()
        reps = 3 if bits > 14 else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return reduce_max((time.perf_counter() - t0) / reps * 1e3)

    def describe(data, prover, ms, what):
        cfg = data.config
        widths = data.num_constants + cfg["num_routed_wires"] + cfg["num_wires"] + 2 * (1 + data.num_partial_products) + 2 * 8
        return {"ms_per_proof": ms, "proofs_per_s": world * 1e3 / ms, "proof_bytes": prover.proof_bytes, "rows": data.n,
                "wires": cfg["num_wires"], "committed_polys": widths, "gate_types": len(data.gates), "circuit": what,
                "stages_ms": {k: round(v, 3) for k, v in prover.last_timings().items()}}

    # ---- a4-a6: the reference's per-signature circuit (crypto/plonky2_ed25519/src/gadgets/eddsa.rs:34-85) with the witnesses of
    # the three real NEAR approval signatures of fixture C1
    t0 = time.perf_counter()
    j = json.load(open(os.path.join(ROOT, "tests", "golden", "ed25519_near_c1_small.json")))
    msg = bytes.fromhex(j["msg"])
    ent = pipe.ed_circuit(len(msg))
    ed_data, ed_prover = ent.data, ent.provers[0]
    t1 = time.perf_counter()
    fills = [E.fill_ecdsa_targets(ent.targets, msg, bytes.fromhex(x["approval"])[2:], bytes.fromhex(x["validator_tail"])[1:33])
             for x in j["entries"]]
    wn, pn = ed_data.generate_witness_native(fills)       # csrc/plonky2_witness.cpp, one host thread per signature
    t2 = time.perf_counter()
    ms = time_proof(ed_prover, wn[0], [int(x) for x in pn[0]], pipe.ctx.stream_ptr(), 18)
    out[ed_name] = describe(ed_data, ed_prover, ms, "reference Ed25519 circuit, real NEAR signature witness")
    out[ed_name]["host_python_untimed"] = {"circuit_build_or_cache_load_s": t1 - t0, "native_witness_s_per_signature": (t2 - t1) / len(fills)}
    c1_proofs = [ed_prover.prove_bytes(wn[k], [int(x) for x in pn[k]]) for k in range(len(fills))]
    if args.cpu_baseline_ed25519 and rank == 0 and world == 1 and not args.no_cpu_baseline:
        # opt-in (1-3 minutes of host time, ~12 GB of host memory): ONE complete proof of the Ed25519 circuit on the host cores
        try:
            from oracle import cport
            threads_ = host_cores()
            cpu_bytes, secs = cport.plonky2_prove(ed_data, wn[0], [int(x) for x in pn[0]], nthreads=threads_)
            out[ed_name]["cpu_baseline_measured"] = {
                "value": 1.0 / secs["proof"], "unit": "proofs/s", "cores": threads_, "kind": "port", "seconds_per_proof": secs["proof"],
                "sample": "ONE complete proof of the reference Ed25519 circuit (2^18 rows x 234 wires, 20 gate types, real NEAR "
                          "signature) with oracle/c/plonky2_prover_oracle.c, C + OpenMP; circuit preprocessing (%.1f s) excluded"
                          % secs["preprocess"],
                "stages_s": {k: round(v, 3) for k, v in secs.items() if k != "threads"},
                "proof_bytes_equal_gpu": cpu_bytes == c1_proofs[0]}
        except Exception as e:
            out[ed_name]["cpu_baseline_error"] = repr(e)[:300]
    del wn
    # the host interpreter keeps one value array per slot and thread (26 M slots x 12 bytes x 3 threads for this circuit): the
    # blocks below generate the Ed25519 witnesses on the GPU, so the cache goes back to the allocator
    zklc_amd._lib.load().zklc_plonky2_witness_release()

    # ---- a7: `recursive_proof` (prove_crypto/recursion.rs:16-97).  The fold of signatures.rs:97-105 uses two circuit shapes --
    # R(ed, ed) for the first step, R(R, ed) for every later one (its common data is a fixed point) -- the closing proof carries
    # sha256(valid_keys) as 32 public inputs (:125-139) and the last recursion runs with Poseidon-BN128 Merkle caps
    # (bin/prove_block.rs:279-287).  The pipeline's own fold / wrap provers are timed (their circuits stay resident for the blocks).
    rp, rpw = pipe.rp, pipe.rpw
    ed3 = [(ent.common, ent.vd, p_) for p_ in c1_proofs]
    shapes_t = {}

    def build_step(name, prover_, *a, **kw):
        t_ = time.perf_counter()
        rc, proof = prover_.recursive_proof(*a, raw=True, **kw)
        shapes_t[name] = (time.perf_counter() - t_, rc)
        return (rc.common, rc.verifier_only, proof)
    r1 = build_step("fold_first_R(ed,ed)", rp, ed3[0], ed3[1])
    r2 = build_step("fold_R(R,ed)", rp, r1, ed3[2])
    assert r2[0] == r1[0], "the fold's common data must be a fixed point"
    pis32 = list(hashlib.sha256(b"bench").digest())
    rf = build_step("closing_R(R)+32PI", rp, r2, None, pis32)
    build_step("wrap_bn128_R(closing)", rpw, rf)
    fold_steps = {"fold_first_R(ed,ed)": (rp, (ed3[0], ed3[1]), {}), "fold_R(R,ed)": (rp, (r1, ed3[2]), {}),
                  "closing_R(R)+32PI": (rp, (r2, None, pis32), {}), "wrap_bn128_R(closing)": (rpw, (rf,), {})}
    for name, (prover_, a, kw) in fold_steps.items():
        build_s, rc = shapes_t[name]
        prover_.recursive_proof(*a, raw=True, **kw)          # second run: circuit resident, program compiled
        host = dict(prover_.last_host_ms)
        wires = rc.wire_buffer()[0].copy()
        pis_ = pis32 if "32PI" in name else []
        ms = time_proof(rc.prover, wires, pis_, prover_.ctx.stream_ptr(), rc.data.degree_bits)
        key = "recursion_%s_2p%dx135" % (name, rc.data.degree_bits)
        out[key] = describe(rc.data, rc.prover, ms, "in-circuit verifier (recursive_proof) over real inner proofs")
        out[key]["host_ms_per_call"] = {k: round(v, 3) for k, v in host.items()}
        out[key]["host_python_untimed"] = {"first_call_s_circuit_build_upload_program": build_s}
        if name == "fold_R(R,ed)" and rank == 0 and world == 1 and not args.no_cpu_baseline:
            # the honest CPU number: ONE complete proof of this circuit (the step the fold repeats per signature) with the
            # oracle's C + OpenMP prover on the host cores, same circuit, same witness; its bytes must be the GPU's
            try:
                from oracle import cport
                threads_ = host_cores()
                gpu_bytes = rc.prover.prove_bytes(wires, pis_)
                cpu_bytes, secs = cport.plonky2_prove(rc.data, wires, pis_, nthreads=threads_)
                out[key]["cpu_baseline"] = {
                    "value": 1.0 / secs["proof"], "unit": "proofs/s", "cores": threads_, "kind": "port",
                    "sample": "ONE complete proof of this circuit (wires commitment, Z / partial products, quotient with all 13 gate "
                              "types, openings, FRI with proof of work and queries) with oracle/c/plonky2_prover_oracle.c, C + OpenMP, "
                              "no SIMD intrinsics; circuit preprocessing (%.2f s) excluded as it is for the GPU" % secs["preprocess"],
                    "stages_s": {k: round(v, 4) for k, v in secs.items() if k != "threads"},
                    "proof_bytes_equal_gpu": cpu_bytes == gpu_bytes,
                    "gpu_speedup": secs["proof"] * 1e3 / ms}
            except Exception as e:      # the baseline is a report: it must not take the bench line down
                out[key]["cpu_baseline_error"] = repr(e)[:300]

    # ---- 8(f).2: the other circuits of a block proof -- SHA-256 (prove_crypto/sha256.rs:62-83; header / bp_hash / valid_keys
    # hashes) and the keys / stakes circuit (prove_block_data/keys_stakes.rs:18-243) on the 100-validator fixture
    from zklc_amd.plonky2 import sha256 as SHA
    from zklc_amd import keys_stakes as KS
    c2f = json.load(open(os.path.join(ROOT, "tests", "golden", "ed25519_near_c2_100.json")))
    vals = [len(e["account_id"]).to_bytes(4, "little") + e["account_id"].encode() + bytes.fromhex(e["validator_tail"]) for e in c2f["entries"]]
    vkeys = b"".join(bytes([pos]) + vals[pos][-48:-16] for pos, e in enumerate(c2f["entries"]) if len(bytes.fromhex(e["approval"])) == 66)
    for name, (data_, pw_) in {
            "sha256_208B_inner_lite": (lambda d_w: (d_w[0], SHA.sha256_witness(d_w[1], bytes(208))))(SHA.sha256_circuit(208)),
            "keys_stakes_100_validators": (lambda d: (d[0], {**{t: x for ts, v in zip(d[1], vals) for t, x in zip(ts, v)},
                                                             **dict(zip(d[2], vkeys))}))(KS.keys_stakes_circuit(vkeys, [len(v) for v in vals]))}.items():
        t_ = time.perf_counter()
        data_.witness_program(list(pw_))
        wn_, pn_ = data_.generate_witness_native([pw_])
        t_w = time.perf_counter() - t_
        pr_ = data_.prover(pipe.fold_ctx, HASH_GL)
        ms = time_proof(pr_, wn_[0], [int(x) for x in pn_[0]], pipe.fold_ctx.stream_ptr(), data_.degree_bits)
        key = "%s_2p%dx135" % (name, data_.degree_bits)
        out[key] = describe(data_, pr_, ms, "the reference's circuit (restated), real witness")
        out[key]["host_python_untimed"] = {"program_compile_and_native_witness_s": t_w}
        pr_.close()

    # ---- FULL Block_i proofs (BASELINE configs[2]: `prove_block_bft`, bft.rs:38-500, the path of bin/prove_random.rs) on the
    # reference's own data: NEAR mainnet blocks 121798939..43 with the 100 block producers of their epoch, Block_0 of the previous
    # epoch and the last block of the one before (tests/golden/block_window_HPi5.json: borsh headers pinned by the block hashes),
    # through zklc_amd.pipeline.BlockPipeline (its docstring lists the threads and streams)
    win = json.load(open(os.path.join(ROOT, "tests", "golden", "block_window_HPi5.json")))
    window = BlockWindow.from_fixture(win)
    want = window.expected_public_inputs()[0]
    n_sig = sum(1 for a in window.blocks[3][0]["approvals"] if len(a) == 66)
    barrier()
    strong_only = args.scaling == "strong" and world > 1
    overlap = not strong_only and not args.no_block_overlap
    # builds (or loads) and uploads the circuits of every shape of the DAG (host Python, one-time); through the SAME entry point the
    # timed blocks use, so that everything it creates lazily (the header thread's context and provers) exists before the clock starts
    if overlap:
        pipe.prove_stream([window])
    else:
        pipe.prove_block_bft(window)
    t_setup = time.perf_counter() - t_setup
    # The circuits are tens of millions of long-lived Python objects: a full generational collection walks all of them and stalled
    # ONE block in ~15 by 15 s (round 2).  Standard remedy for a long-running service: collect once, then move everything alive to
    # the permanent generation, so that later collections only look at what a block allocates.
    import gc
    gc.collect()
    try:                                   # what the builders and the per-circuit stages freed goes back to the OS, not only to the heap
        import ctypes
        ctypes.CDLL("libc.so.6").malloc_trim(0)
    except (OSError, AttributeError):
        pass
    gc.freeze()
    if overlap and args.warmup > 1:
        pipe.prove_stream([window] * (args.warmup - 1))
    else:
        for _ in range(max(0, args.warmup - 1)):
            pipe.prove_block_bft(window, strong=strong_only)
    barrier()
    tele = GpuTelemetry(pci=GpuTelemetry.own_pci())
    if not tele.cards:                       # sysfs does not know the address torch reports: fall back to the busiest card
        tele.close()
        tele = GpuTelemetry()
    tele.mark()
    rss0 = rss_mb()
    steps = max(1, args.steps)

    def digests(r):
        return (hashlib.sha256(r.wrap[1]).hexdigest(), hashlib.sha256(json.dumps(r.block[2], sort_keys=True).encode()).hexdigest())
    tele_steps = []

    def block_done(r):
        tele_steps.append(dict(tele.mark(), rss_mb=rss_mb()))
    import resource

    def clocks_ns():
        # the timed region in every clock a profiler may stamp kernels with: tools/block_accounting.py picks the one the trace uses
        return {n: time.clock_gettime_ns(getattr(time, c)) for n, c in (("monotonic", "CLOCK_MONOTONIC"), ("boottime", "CLOCK_BOOTTIME"),
                                                                       ("realtime", "CLOCK_REALTIME"), ("monotonic_raw", "CLOCK_MONOTONIC_RAW"))}
    def host_load():
        # is the HOST ours?  (the GPU boxes share their CPUs between tenants: a loaded host shows up as block-time noise)
        try:
            la = open("/proc/loadavg").read().split()[:3]
            return {"loadavg": [float(x) for x in la], "cpus_allowed": len(os.sched_getaffinity(0)), "cpus_online": os.cpu_count()}
        except (OSError, ValueError):
            return {}
    # The collector's pauses stop EVERY Python thread of the pipeline (the streams then drain: GPU idle): measured here (gc.callbacks);
    # prove_stream replaces the automatic collections by one young-generation collection per block (ZKLC_STREAM_GC=auto: A/B)
    gc_mode = os.environ.get("ZKLC_STREAM_GC", "block") if overlap else "auto"
    gc_stat = {"n": [0, 0, 0], "s": [0.0, 0.0, 0.0], "max_s": 0.0, "t0": 0.0}

    def gc_cb(phase, info):
        if phase == "start":
            gc_stat["t0"] = time.perf_counter()
        else:
            dt_, g_ = time.perf_counter() - gc_stat["t0"], min(2, int(info.get("generation", 0)))
            gc_stat["n"][g_] += 1
            gc_stat["s"][g_] += dt_
            gc_stat["max_s"] = max(gc_stat["max_s"], dt_)
    gc.callbacks.append(gc_cb)
    # interpreter-wide stalls, whatever their cause (collector, a C call that keeps the GIL, a descheduled process on a shared host):
    # a thread that asks for 2 ms of sleep and records by how much it overslept
    stall = {"late_s": 0.0, "n5": 0, "n20": 0, "n100": 0, "max_s": 0.0, "stop": False}

    def stall_probe():
        t_ = time.perf_counter()
        while not stall["stop"]:
            time.sleep(0.002)
            t2_ = time.perf_counter()
            late = t2_ - t_ - 0.002
            t_ = t2_
            if late > 0.005:
                stall["late_s"] += late
                stall["n5"] += 1
                stall["n20"] += late > 0.02
                stall["n100"] += late > 0.1
                stall["max_s"] = max(stall["max_s"], late)
    stall_th = threading.Thread(target=stall_probe, daemon=True)
    stall_th.start()
    # diagnostic (ZKLC_BENCH_SAMPLE=<file>): every 20 ms where each pipeline thread is (innermost frame inside this repo), written as
    # JSON lines [t, {thread: "file:line"}] -- which worker waits where while a prover queue of the kernel trace is idle
    sampler = {"stop": False, "rows": []}

    def sample_threads():
        names = {}
        while not sampler["stop"]:
            time.sleep(0.02)
            for th in threading.enumerate():
                names[th.ident] = th.name
            row = {}
            for ident, fr in sys._current_frames().items():
                nm = names.get(ident, "")
                if not nm.startswith("zklc-") and nm != "MainThread":
                    continue
                inner = "%s:%d" % (os.path.basename(fr.f_code.co_filename), fr.f_lineno)
                f2 = fr
                while f2 is not None and ROOT not in f2.f_code.co_filename:
                    f2 = f2.f_back
                ours = "%s:%d" % (os.path.basename(f2.f_code.co_filename), f2.f_lineno) if f2 is not None else "?"
                row[nm] = ours if ours == inner else ours + "<" + inner
            sampler["rows"].append([round(time.perf_counter() - t_all, 3), row])
    sample_th = None
    if os.environ.get("ZKLC_BENCH_SAMPLE"):
        sample_th = threading.Thread(target=sample_threads, daemon=True)
    load0 = host_load()
    clk0 = clocks_ns()
    cpu_all = time.process_time()
    t_all = time.perf_counter()
    if sample_th is not None:
        sample_th.start()
    if overlap:
        # exactly K complete Block_i proofs; consecutive blocks overlap by the tail of the earlier one (BlockPipeline.prove_stream)
        res_list = pipe.prove_stream([window] * steps, block_done)
    else:
        res_list = []
        for _ in range(steps):
            r = pipe.prove_block_bft(window, strong=strong_only)
            if r is not None:
                res_list.append(r)
                block_done(r)
    barrier()
    total_s = reduce_max(time.perf_counter() - t_all)
    gc.callbacks.remove(gc_cb)
    stall["stop"] = True
    stall_th.join()
    if sample_th is not None:
        sampler["stop"] = True
        sample_th.join()
        with open(os.environ["ZKLC_BENCH_SAMPLE"], "w") as f_:
            for row in sampler["rows"]:
                f_.write(json.dumps(row) + "\n")
    clk1 = clocks_ns()
    cpu_all = time.process_time() - cpu_all        # user + system seconds of THIS rank's process (all threads) over the timed blocks
    block_s = total_s / steps
    tele.close()
    tele_all = {k: round(sum(t[k] for t in tele_steps if k in t) / max(1, sum(1 for t in tele_steps if k in t)), 1)
                for k in ("sclk_mhz", "power_w", "busy_pct", "temp_c", "mclk_mhz") if any(k in t for t in tele_steps)}
    if strong_only and rank != 0:
        pipe.close()
        return out            # rank 0 holds the block proofs, verifies them and reports
    # ---- checker role, untimed.  EVERY timed block: public inputs as expected and proof bytes identical (the prover is
    # deterministic: same window -> same bytes) to the LAST block's, whose Block_i proof and Poseidon-BN128 wrap -- the only two
    # proofs of the DAG that no later recursion witness checks -- are verified by the oracle's verifier restatement (pinned by the
    # reference's golden proofs; prove_crypto/recursion.rs:53,81 verifies every inner proof natively)
    from oracle import plonky2_verifier as V
    t_v = time.perf_counter()
    lastr = res_list[-1]
    d_last = digests(lastr)
    same = [digests(r) == d_last and r.block[2]["public_inputs"] == want for r in res_list]
    assert all(same), "a timed block's final proofs differ from the verified one: %s" % same
    V.verify(json.loads(json.dumps(lastr.block[2])), lastr.block[1], lastr.block[0])
    wrc, wraw = lastr.wrap
    wrap_json = S.proof_from_bytes(wraw, wrc.common, HASH_BN128)
    V.verify(json.loads(json.dumps(wrap_json)), wrc.verifier_only, wrc.common)
    assert wrap_json["public_inputs"] == want, "wrap proof public inputs"
    t_v = time.perf_counter() - t_v
    per_step_s = [round(b.t_done - a.t_done, 4) for a, b in zip(res_list, res_list[1:])]
    per_step_s = [round(res_list[0].t_done - t_all, 4)] + per_step_s
    out["block_i"] = {"value": (1 if strong_only else world) / block_s, "unit": "proofs/s", "seconds_per_block": block_s,
                      "blocks_timed": steps, "scaling": "strong" if strong_only else "weak",
                      "per_step_s": per_step_s, "latency_s": [round(r.t_done - r.t0, 4) for r in res_list],
                      "telemetry_mean": tele_all, "per_step_telemetry": tele_steps, "rss_mb_before": rss0, "rss_mb_after": rss_mb(),
                      "blocks_overlapped": overlap, "keys_stakes_cache_hits": pipe.ks_prover.cache_hits,
                      "seconds_until_signature_aggregate": lastr.t_signatures - lastr.t0, "streams": pipe.nthreads + 2 + (1 if pipe.hprover is not None else 0),
                      "approvals": n_sig, "witness_on": "gpu" if pipe.dev_wit else "host", "witness_batch": pipe.wchunk,
                      "witness_producer_seconds": lastr.witness_s, "preverify_ms": lastr.t_verify * 1e3,
                      "fold_thread_seconds": {k: round(v / 1e3, 3) for k, v in lastr.fold_host_ms.items()},
                      "dag_thread_seconds": {k: round(v, 3) for k, v in lastr.dag_seconds.items()},
                      "dag_thread_counts": dict(lastr.dag_counts), "keys_stakes_thread_seconds": lastr.keys_stakes_s,
                      "wrap_proof_bytes": len(wraw), "blocks_checked": len(res_list),
                      "final_proof_verified": True, "final_proof_verify_s_untimed": t_v,
                      "first_block_s_incl_circuit_construction": t_setup,
                      "host_cpu_s_per_block": cpu_all / steps, "host_cores_busy": cpu_all / max(total_s, 1e-9),
                      "peak_rss_mb": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0,
                      "timed_region_clock_ns": {k: [clk0[k], clk1[k]] for k in clk0},
                      "host_load_before": load0, "host_load_after": host_load(),
                      "hbm_used_gb": round((torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9, 1),   # the whole device, all processes
                      "witness_buffers": pipe.nbuf, "device_share": pipe.device_share,
                      "gc": {"mode": gc_mode, "switch_interval_ms": os.environ.get("ZKLC_SWITCH_INTERVAL_MS", "0.5") if overlap else None, "collections_by_generation": gc_stat["n"], "pause_s_by_generation": [round(x, 3) for x in gc_stat["s"]],
                             "pause_s_per_block": round(sum(gc_stat["s"]) / steps, 4), "max_pause_s": round(gc_stat["max_s"], 4)},
                      "interpreter_stalls": {"late_s_per_block": round(stall["late_s"] / steps, 4), "over_5ms": stall["n5"], "over_20ms": int(stall["n20"]),
                                             "over_100ms": int(stall["n100"]), "max_s": round(stall["max_s"], 4),
                                             "note": "a 2 ms sleeper's oversleep beyond 5 ms, summed: time in which no Python thread of the pipeline could run"}}
    if world > 1 and not strong_only and not args.no_strong_section:
        # the STRONG form of the block measured in the same run, so that one SCALE run holds both: all ranks prove
        # ONE block per step (signature shards, local folds, a binary-tree fold over the ranks, the header proofs on the other ranks,
        # the joins and the wrap on rank 0).  One untimed block builds the tree fold's circuit shapes.
        pipe.prove_block_bft(window, strong=True)
        barrier()
        ns = max(2, steps // 4)
        t_s = time.perf_counter()
        for _ in range(ns):
            sres = pipe.prove_block_bft(window, strong=True)
        barrier()
        strong_s = reduce_max(time.perf_counter() - t_s) / ns
        if rank == 0:
            assert sres.block[2]["public_inputs"] == want, "strong mode: block proof public inputs"
            sw_rc, sw_raw = sres.wrap
            V.verify(json.loads(json.dumps(S.proof_from_bytes(sw_raw, sw_rc.common, HASH_BN128))), sw_rc.verifier_only, sw_rc.common)
            out["block_i"]["strong"] = {"value": 1.0 / strong_s, "unit": "proofs/s", "seconds_per_block": strong_s, "blocks_timed": ns,
                                        "scaling": "strong", "final_proof_verified": True,
                                        "speedup_vs_one_rank_weak_block": block_s / strong_s}
    if args.c5_validators and not strong_only:
        try:
            # BASELINE configs[4] (C5), the part the unmodified circuits can express: a synthetic epoch of N validators who all sign the
            # same Approval message -> batched GPU pre-verification, N Ed25519-circuit proofs (witnesses on the GPU), the left fold and
            # the closing proof with sha256(valid_keys), through BlockPipeline.prove_approvals (one rank = one GPU)
            from oracle import ed25519_ref as ref
            nv = int(args.c5_validators)
            if nv < 0:
                nv = 1000 if time.perf_counter() - T_PROCESS_START < args.c5_budget_s else 200
            c2_msg = window.approval_sets()[0][0]
            keys = [ref.synthetic_seed(1, i) for i in range(nv)]
            vals5 = [b"\x04\x00\x00\x00test\x00" + ref.keypair(k_)[2] + (10**30 + i).to_bytes(16, "little") for i, k_ in enumerate(keys)]
            apps5 = [b"\x01\x00" + ref.sign(k_, c2_msg) for k_ in keys]
            t_ = time.perf_counter()
            (rc5, proof5), vk5 = pipe.prove_approvals(c2_msg, apps5, vals5)
            dt5 = reduce_max(time.perf_counter() - t_)
            assert len(vk5) == 33 * nv
            V.verify(json.loads(json.dumps(S.proof_from_bytes(proof5, rc5.common, HASH_GL))), rc5.verifier_only, rc5.common)
            out["c5_synthetic_epoch"] = {"validators": nv, "seconds": dt5, "signature_proofs_per_s": world * nv / dt5,
                                         "preverify_ms": pipe.last.t_verify * 1e3, "aggregate_verified": True,
                                         "witness_on": "gpu" if pipe.dev_wit else "host",
                                         "note": "N Ed25519-circuit proofs + %d fold steps + closing proof on one GPU per rank; the keys / "
                                                 "stakes circuit of the reference cannot express N > 255 positions (pos as u8), so C5 stops "
                                                 "at the signature aggregate" % (nv - 1)}
        except Exception as e:      # a secondary figure: never a reason to lose the line (the headline is already measured)
            if world > 1:
                raise                   # the other ranks are in a collective: fail the job rather than leave them waiting
            out["c5_synthetic_epoch_error"] = repr(e)[:300]
    pipe.close()
    return out


def r3(x, nd=4):
    """numbers of the compact line: `nd` significant digits"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    return x


def compact_line(full):
    """The ONE stdout line of the driver contract, kept below 8 KB (round 3's 20 KB line was not parsed by the driver): the
    contract keys, `roofline` (+ `valu`), `cpu_baseline` and one short entry per stage; everything else is in the detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "final_proof_verified")
    line = {k: full[k] for k in keep if k in full}

    def roof(r):
        if not r:
            return None
        o = {k: r3(r.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms") if k in r}
        if "valu" in r:
            o["valu"] = {k: r3(r["valu"][k]) for k in ("bound", "achieved", "peak", "unit", "frac", "ubench") if k in r["valu"]}
        if "block" in r:
            o["block"] = {k: r3(r["block"][k]) for k in ("valu_lane_instr_per_block", "frac_of_issue_limit", "sum_exclusive_kernel_s",
                                                         "wall_s", "sclk_mhz", "gpu_idle_s", "floor_s_at_issue_limit") if r["block"].get(k) is not None}
        return o
    line["roofline"] = roof(full.get("roofline"))
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: r3(cb[k]) for k in ("value", "unit", "cores", "kind", "sample", "seconds_per_block", "gpu_speedup") if k in cb}
    st = full.get("stages") or {}
    cs = {}
    # `lde` of the line = the PRODUCT shape 234 x (2^18 -> 2^21) every commitment of the Ed25519 circuit extends; C3's own shape
    # (2^17 -> 2^20, BASELINE configs[2]) rides beside it as `lde_c3` (round 6; the detail file keeps them as lde_2p18 / lde)
    have18 = bool(st.get("lde_2p18") and "value" in st["lde_2p18"])
    for name in ("msm", "lde", "lde_2p18", "merkle", "ed25519_verify"):
        s_ = st.get(name)
        if name == "lde" and have18:
            name = "lde_c3"
        elif name == "lde_2p18" and have18:
            name = "lde"
        if s_ and "value" in s_:
            e = {"value": r3(s_["value"]), "unit": s_["unit"]}
            if "ms" in s_:
                e["ms"] = r3(s_["ms"])
            r = s_.get("roofline") or {}
            if "frac" in r:
                e["hbm_frac"] = r3(r["frac"])
            if r.get("traffic"):
                e["traffic"] = r3(float(r["traffic"]))
            if "valu" in r:
                e["valu_frac"] = r3(r["valu"]["frac"])
            if "cpu_baseline" in s_:
                e["cpu"] = r3(s_["cpu_baseline"]["value"])
            if "distributions" in s_:
                e["dist_ms"] = {k: r3(v["ms"]) for k, v in s_["distributions"].items()}
            if "fixed_base" in s_ and "value" in s_["fixed_base"]:
                fb = s_["fixed_base"]
                e["fixed_base"] = {"value": r3(fb["value"]), "ms": r3(fb["ms"]), "W_ms": r3(fb.get("witness_like_ms")), "equal": fb["equals_plain_result"]}
            if "strong" in s_:
                e["strong"] = {"value": r3(s_["strong"]["value"]), "ms": r3(s_["strong"]["ms"]), "equal": s_["strong"].get("equals_single_gpu_result")}
            cs[name] = e
    for name, s_ in (st.get("bn254_extras") or {}).items():
        cs[name] = {"value": r3(s_["value"]), "unit": s_["unit"].split(" (")[0], "ms": r3(s_["ms"])}
    pr = st.get("prove") or {}
    if pr:
        cs["prove_ms"] = {k.split("_2p")[0]: r3(v["ms_per_proof"]) for k, v in pr.items() if isinstance(v, dict) and "ms_per_proof" in v}
        if "c5_synthetic_epoch" in pr:
            c5 = pr["c5_synthetic_epoch"]
            cs["c5"] = {"validators": c5["validators"], "seconds": r3(c5["seconds"]), "sig_proofs_per_s": r3(c5["signature_proofs_per_s"])}
    if cs:
        line["stages"] = cs
    blk = full.get("block_i")
    if blk:
        b = {k: r3(blk[k]) for k in ("seconds_per_block", "blocks_timed", "blocks_checked", "blocks_overlapped", "approvals", "streams",
                                     "witness_on", "first_block_s_incl_circuit_construction", "rss_mb_after", "host_cpu_s_per_block",
                                     "host_cores_busy", "peak_rss_mb") if k in blk}
        b["per_step_s"] = [r3(x, 3) for x in blk.get("per_step_s", [])][:64]
        tm = blk.get("telemetry_mean") or {}
        b["gpu"] = {k: tm[k] for k in ("sclk_mhz", "power_w", "busy_pct", "temp_c") if k in tm}
        if "strong" in blk:
            b["strong"] = {k: r3(blk["strong"][k]) for k in ("value", "seconds_per_block", "blocks_timed", "final_proof_verified",
                                                             "speedup_vs_one_rank_weak_block")}
        line["block_i"] = b
    line["detail"] = full.get("detail")
    return line


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start the N ranks (one per GPU, RCCL over xGMI) and
    become the launcher -- rank 0 of the children prints the line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed steps; a step = one full Block_i proof (with --no-prove / --no-stages: one "
                                                         "launch of the batched Ed25519 verification)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed steps (the first block also builds and uploads every circuit)")
    ap.add_argument("--verify-steps", type=int, default=20, help="timed launches of the Ed25519 verification stage")
    ap.add_argument("--verify-warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=8192, help="Block_i approval sets per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-ed25519", action="store_true", default=True,
                    help="(default) time ONE complete CPU proof of the Ed25519 circuit with the oracle's C prover on the host cores "
                         "(~40 s + 10 s preprocessing on 16 cores, ~12 GB of host memory): the block's CPU baseline is then MEASURED")
    ap.add_argument("--no-cpu-baseline-ed25519", dest="cpu_baseline_ed25519", action="store_false",
                    help="skip that proof: the Ed25519 shape's CPU time is then the measured wires commitment, the other stages scaled")
    ap.add_argument("--no-stages", action="store_true", help="only the headline C2 measurement")
    ap.add_argument("--msm-log", type=int, default=22, help="log2 of the MSM size per GPU")
    ap.add_argument("--no-prove", action="store_true", help="skip the plonky2 proof stage")
    ap.add_argument("--no-bn254-extras", action="store_true", help="skip the G2 MSM / Fr NTT / pairing stage")
    ap.add_argument("--prove-streams", type=int, default=4, help="proofs in flight per GPU in the Block_i stage")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank proves its own block per step; strong: all ranks prove ONE block per step (signature shards, "
                         "tree fold over the ranks, header proofs on the other ranks)")
    ap.add_argument("--no-block-overlap", action="store_true", help="prove the timed blocks strictly one after the other (no overlap of "
                    "a block's tail with the next block's signature proofs)")
    ap.add_argument("--no-strong-section", action="store_true", help="multi-GPU runs: skip the extra strong-scaling blocks after the weak region")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo lets several ranks share one GPU)")
    ap.add_argument("--c5-validators", type=int, default=-1, help="the C5 stage: a synthetic epoch of this many validators through "
                    "BlockPipeline.prove_approvals, reported under stages.prove.c5_synthetic_epoch (1000 = BASELINE configs[4], ~63 s; "
                    "200: ~13 s; 0 skips it).  Default -1 = automatic: 1000 when the run is younger than --c5-budget-s seconds at that "
                    "point, else 200 -- the line says which (c5.validators)")
    ap.add_argument("--c5-budget-s", type=float, default=330.0, help="with --c5-validators -1: run BASELINE's 1000 validators only if "
                    "fewer than this many seconds have passed since the process started (the driver's command has spent ~280 s by then)")
    ap.add_argument("--extra-stages", action="store_true", help="also time the fixed-base form of the 2^22 G1 multi-exponentiation (table of "
                    "the bases built once; 4.3 GB) and the coset LDE at the Ed25519 circuit's shape 234 x (2^18 -> 2^21); both have "
                    "their own profiles under profiles/ (r05g_msm_fixed_base_*, r05a_lde_*)")
    ap.add_argument("--host-witness", action="store_true", help="Ed25519-circuit witnesses from the host interpreter (threads + PCIe) instead of the GPU")
    ap.add_argument("--witness-batch", type=int, default=32, help="signatures per device witness batch (<= 64; 0.5 GB of HBM each)")
    ap.add_argument("--detail", default=os.environ.get("ZKLC_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json")),
                    help="file for the full report (per-circuit stages, telemetry, notes); the stdout line names it")
    ap.add_argument("--cpu-only", action="store_true", help="functional run of the launcher / collectives / line format without a GPU "
                    "(CPU tests): no kernels are launched and the line says so; never a measurement")
    args = ap.parse_args()

    # a bench that does not finish says WHERE it is: every 20 minutes the stacks of all threads go to stderr (diagnosis only)
    import faulthandler
    faulthandler.dump_traceback_later(1200, repeat=True, file=sys.stderr)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        self_launch(args)                 # does not return
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    backend = args.backend or ("gloo" if args.cpu_only else "nccl")
    if args.cpu_only:
        return main_cpu_only(args, rank, world, backend)
    import zklc_amd
    local_rank %= max(1, torch.cuda.device_count())       # --backend gloo: several ranks may share one GPU (functional runs)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    # the circuits of the block DAG are built once per MACHINE: the ranks of a multi-GPU run (and the driver's N = 1, 2, 4, 8 runs one
    # after the other) load what an earlier process built (zklc_amd/plonky2/circuit_cache.py; ZKLC_CIRCUIT_CACHE= (empty) disables)
    os.environ.setdefault("ZKLC_CIRCUIT_CACHE", os.path.join(ROOT, ".circuit_cache"))

    pk, sg, ms = make_base_set()
    base_n = pk.shape[0]
    reps = (args.blocks + BASE_BLOCKS - 1) // BASE_BLOCKS
    n = reps * base_n
    dev = torch.device("cuda", local_rank)
    d_pk = torch.from_numpy(pk).to(dev).repeat(reps, 1).contiguous()
    d_sg = torch.from_numpy(sg).to(dev).repeat(reps, 1).contiguous()
    d_ms = torch.from_numpy(ms).to(dev).repeat(reps, 1).contiguous()
    # 1 % corrupted: flip one bit of signature i (i % 100 == rank-dependent offset)
    bad = torch.arange((7 + rank) % 100, n, 100, device=dev)
    d_sg[bad, (bad % 64)] ^= 1
    d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    expect = torch.ones(n, dtype=torch.uint8, device=dev)
    expect[bad] = 0

    ctx = zklc_amd.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    stream.wait_stream(torch.cuda.current_stream())

    def step():
        ctx.ed25519_verify_batch_dev(d_pk, d_sg, d_ms, MSG_LEN, MSG_STRIDE, n, d_ok, stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    headline_verify = args.no_stages or args.no_prove      # without the prove stage the Ed25519 launches are the steps
    v_steps = args.steps if headline_verify else args.verify_steps
    v_warm = args.warmup if headline_verify else args.verify_warmup
    for _ in range(v_warm):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(v_steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, v_steps)

    # results are checked after the timed region (all GPUs)
    assert torch.equal(d_ok, expect), "rank %d: GPU bitmap differs from the expected validity pattern" % rank
    n_valid = int(d_ok.sum())
    if world > 1:
        cdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, kernel_ms], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
        c = torch.tensor([n_valid], device=cdev, dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        n_valid = int(c[0])

    stages = None
    if not args.no_stages:
        del d_pk, d_sg, d_ms, d_ok, expect
        torch.cuda.empty_cache()
        stages = run_stages(args, ctx, dev, stream, rank, world, with_cpu=(world == 1 and not args.no_cpu_baseline))

    if rank == 0:
        total = n * world
        value = total * v_steps / elapsed
        achieved = BYTES_PER_SIG * n / (kernel_ms * 1e-3) / 1e9
        verify = {
            "metric": "Block_i approval-signature verifications/s (100-validator Ed25519 batch, stage (a) of the "
                      "BFT-finality proof path)",
            "value": value, "unit": "sig/s", "n_gpus": world, "steps": v_steps, "warmup": v_warm,
            "ms_per_step": elapsed / v_steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "C2: batched Ed25519 verify, %d Block_i approval sets x %d validators per GPU per step "
                                   "(%d signatures, 41-byte per-block message, 1%% corrupted)" % (n // VALIDATORS, VALIDATORS, n),
                       "signatures_per_gpu": n, "blocks_per_s": value / VALIDATORS, "valid": n_valid,
                       "kernel": "ed25519_verify_kernel_v1 (8-entry per-lane table in LDS, radix-2^25.5 field)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(n),
                         "kernel": "ed25519_verify_kernel", "kernel_ms": kernel_ms,
                         "note": "algorithmic bytes = 97 B/signature; traffic = HBM bytes per launch from "
                                 "profiles/ed25519_pmc_latest.json (2*FETCH_SIZE + WRITE_SIZE); the kernel is "
                                 "integer-VALU-bound (~6e5 lane-instructions per signature), see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            verify["cpu_baseline"] = cpu_baseline(pk, sg, ms)
        blk = (stages or {}).get("prove", {}).get("block_i")
        if blk is None:
            out = verify
            if stages is not None:
                out["stages"] = stages
        else:
            # BASELINE.json's metric, on its configs[2]: one step = one full Block_i BFT-finality proof
            mk = stages["merkle"]
            edp = stages["prove"]["ed25519_circuit_2p18x234"]
            out = {
                "metric": "Block_i BFT-finality proofs/sec (100 validators)", "value": blk["value"], "unit": "proofs/s",
                "n_gpus": world, "steps": blk["blocks_timed"], "warmup": max(1, args.warmup),
                "ms_per_step": blk["seconds_per_block"] * 1e3, "higher_is_better": True, "scaling": blk["scaling"], "vs_baseline": None,
                "dtype": "u64", "data": "NEAR mainnet block window shipped with the reference (tests/golden/block_window_HPi5.json)",
                "config": {"workload": "configs[2]: full plonky2 BFT-finality proof of Block_i (prove_block_bft, 5-block window, 100 "
                                       "validators, %d approvals), one block per GPU per step, end to end from the borsh bytes"
                                       % blk["approvals"],
                           "approvals": blk["approvals"],
                           "proofs_per_block": dict(blk["dag_thread_counts"], ed25519_circuit=blk["approvals"],
                                                    fold_and_closing_recursions=blk["approvals"], keys_stakes_and_its_hash=3, bn128_wrap=1),
                           "msm_2p22_melem_per_s": r3(stages["msm"]["value"])},
                "roofline": dict(mk["roofline"], kernel="gl_hash_leaves_kernel (Poseidon leaf hashing: 53 % of one proof stream's kernel time, ~20 % of the "
                                        "summed kernel time of the overlapped block trace, the largest single kernel in both)",
                                 kernel_ms=mk["ms"]),
                "final_proof_verified": blk["final_proof_verified"],
                "block_i": blk, "stages": dict(stages, ed25519_verify=verify),
            }
            pm = poseidon_pmc()
            if pm is not None:
                out["roofline"]["traffic"] = pm.get("hbm_bytes_per_launch")
                if "SQ_INSTS_VALU_per_launch" in pm:
                    out["roofline"]["valu"] = valu_block(pm["SQ_INSTS_VALU_per_launch"], mk["ms"], "gl_hash_leaves_kernel, "
                                                         "profiles/poseidon_pmc_latest.json")
                    out["roofline"]["valu"]["instructions_per_permutation"] = pm["SQ_INSTS_VALU_per_launch"] * 64 / ((1 << 20) * 30)
            # the HEADLINE's own fraction: VALU lane-instructions ONE block executes -- SQ_INSTS_VALU summed over
            # the ~28.7 k dispatches of a timed block, a rocprofv3 PMC pass of this very command (tools/gpu_block_accounting.sh ->
            # profiles/block_pmc_latest.json; the prover is deterministic, the count does not depend on the run) -- over the LIVE block
            # time x the SIMDs' multi-pass issue limit.  Valid for the window the count was taken on (73 approvals) and one rank's block.
            bp = pmc_json("block_pmc_latest.json")
            if bp is not None and blk["approvals"] == 73 and blk["scaling"] == "weak":
                lanes = float(bp["valu_lane_instr_per_block"])
                tr = bp.get("trace", {})
                out["roofline"]["block"] = {
                    "valu_lane_instr_per_block": lanes, "wall_s": blk["seconds_per_block"],
                    "frac_of_issue_limit": lanes / (blk["seconds_per_block"] * VALU_INT_PEAK_TLOPS * 1e12),
                    "floor_s_at_issue_limit": lanes / (VALU_INT_PEAK_TLOPS * 1e12),
                    "sum_exclusive_kernel_s": bp.get("sum_serialised_kernel_s"), "gpu_idle_s": tr.get("idle_s"),
                    "sclk_mhz": (blk.get("telemetry_mean") or {}).get("sclk_mhz"),
                    "note": "frac = lane-instructions / (live wall x 39.3 T/s); sum_exclusive_kernel_s = the block's kernels run one at a time "
                            "(their own durations under counter collection); gpu_idle_s = time with no kernel resident in the kernel trace of "
                            "the overlapped run (%s)" % str(bp.get("source", "profiles/block_pmc_latest.json"))[:60]}
            try:
                if "cpu_baseline" in edp:
                    cb = edp["cpu_baseline"]
                    fold = next((v["cpu_baseline"] for k, v in stages["prove"].items()
                                 if k.startswith("recursion_fold_R(R,ed)") and isinstance(v, dict) and "cpu_baseline" in v), None)
                    cnt = blk["dag_thread_counts"]      # a header-hash chain is 4 SHA-256 proofs + 5 recursions (header_bphash.py)
                    n_small = (9 * cnt.get("prove_header_hash", 0) + cnt.get("prove_eq_array", 0) + cnt.get("recursive_proof", 0)
                               + cnt.get("prove_bp_hash", 0) + blk["approvals"] + 3 + 1)
                    if fold is not None:
                        fold_s = 1.0 / fold["value"]
                        block_s = blk["approvals"] * cb["seconds_per_proof"] + n_small * fold_s
                        out["cpu_baseline"] = {
                            "value": 1.0 / block_s, "unit": "proofs/s", "cores": cb["cores"], "kind": "port",
                            "sample": "oracle C+OpenMP prover: one complete proof at the fold shape measured (%.2f s, bytes == GPU: %s); "
                                      "Ed25519 shape: %s (%.1f s/proof); block = %d x that + %d small proofs at the fold-shape time"
                                      % (fold_s, fold.get("proof_bytes_equal_gpu"),
                                         "measured whole" if "cpu_baseline_measured" in edp else "wires commitment measured, other stages scaled by the fold shape's ratio",
                                         cb["seconds_per_proof"], blk["approvals"], n_small),
                            "seconds_per_block": block_s, "gpu_speedup": block_s / blk["seconds_per_block"]}
                    else:
                        out["cpu_baseline"] = {"value": cb["value"] / blk["approvals"], "unit": "proofs/s (upper bound)", "cores": cb["cores"],
                                               "kind": "port",
                                               "sample": "wires commitment (coset LDE + Poseidon Merkle tree) of ONE of the %d Ed25519-circuit "
                                                         "proofs of a block, oracle/c/goldilocks_oracle.c, bounded samples" % blk["approvals"]}
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "error": repr(e)[:300]}
            del out["stages"]["prove"]["block_i"]
        emit(out, args.detail)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def emit(out, detail_path):
    """full report -> the detail file; ONE compact JSON line -> stdout"""
    try:
        with open(detail_path, "w") as f:
            json.dump(out, f, indent=1)
        out["detail"] = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path
    except OSError as e:
        out["detail"] = "not written: %r" % (e,)
    line = json.dumps(compact_line(out), separators=(",", ":"))
    assert len(line) < 8192, "bench line too long for the driver (%d bytes)" % len(line)
    print(line, flush=True)


def main_cpu_only(args, rank, world, backend):
    """`--cpu-only`: the launcher, the process group, the collectives of the strong MSM section (72-byte partials, all-gather) and
    the line format, with NO kernel launched -- the CPU test of `python bench.py --gpus 2` (tests/test_bench_contract.py)."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend)
    part = torch.zeros(9, dtype=torch.int64)
    part[0] = rank + 1
    got = [torch.zeros(9, dtype=torch.int64) for _ in range(world)]
    if world > 1:
        dist.all_gather(got, part)
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t[0]) == world - 1
    else:
        got = [part]
    assert [int(g[0]) for g in got] == list(range(1, world + 1))
    if rank == 0:
        emit({"metric": "Block_i BFT-finality proofs/sec (100 validators)", "value": None, "unit": "proofs/s", "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": "u64", "data": "none (--cpu-only: functional run of the launcher, no GPU work, not a measurement)",
              "config": {"workload": "none", "ranks_seen": len(got)}}, args.detail)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
