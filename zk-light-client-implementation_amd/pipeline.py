"""`prove_block_bft` as a pipeline on one MI355X (and, in the strong form, over the GPUs of a node).

Reference: near_bft_finality/src/prove_bft/bft.rs:38-500 (`prove_block_bft`), prove_block_data/signatures.rs:43-141
(`prove_approvals`; :144-274 is its NATS fan-out, which this replaces on a node), bin/prove_block.rs:279-287 (the BN128 wrap).
`prove_bft.BlockProver` is the reference's sequential driver restated; this module runs the SAME nodes of the SAME DAG -- the
proofs are byte-identical to the sequential driver's (tests/test_gpu_stream_pipeline.py) -- on concurrent host threads, each with its own
zklc context (= HIP stream), so that the GPU always has several proofs in flight:

  signature stage (per approval set)
    a3  one batched Ed25519 pre-verification launch (signatures.rs:79)
    a5  witness producer: the generator program of the Ed25519 circuit on the GPU for a batch of signatures
        (csrc/plonky2_witness_dev.hip), wire matrices written in HBM, double-buffered (`host_witness=True`: host interpreter
        threads + pinned buffers)
    a6  `prove_streams - 1` Ed25519 provers, each with a resident copy of the circuit on its own stream
  fold / DAG stage
    a7  fold thread: agg = recursive_proof(agg, sig_i) as soon as signature proof i exists (signatures.rs:97-105), then the
        closing proof with sha256(valid_keys) (:125-139); high-priority stream
    8f  keys / stakes thread (needs only valid_keys, known after the pre-check) and the DAG thread (`BlockProver` over a stub
        `prove_approvals` that waits for the fold thread: header-hash chains, bp_hash, heights, equalities, the joining recursions,
        the Poseidon-BN128 wrap)
  consecutive blocks (`prove_stream`): the signature stage of block b + 1 starts when the last signature proof of block b is out;
  the tail of block b (last fold steps, closing proof, joins, wrap) completes beside it.

Strong form (`strong=True`, world > 1): contiguous shards of every approval set's signature proofs, local left folds, a
binary-tree fold of the partial aggregates over the ranks (distributed.tree_fold), header proofs and keys / stakes on the other
ranks, joins and wrap on rank 0.  No data-path collective in the weak form (every rank its own blocks).
"""
import hashlib
import os
import sys
import queue
import threading
import time
from concurrent.futures import Future

import numpy as np

from . import distributed as DIST
from . import signatures as SG


class BlockWindow:
    """The arguments of `prove_block_bft` (bft.rs:38-62): two or three epoch-boundary blocks as (borsh bytes, hash), the window
    [(fields, borsh bytes)] in the reference's order ([Block_i+4 .. Block_i] or [Block_4 .. Block_0, Block_n-1]) and the
    validator lists as borsh ValidatorStake bytes."""

    def __init__(self, ep2_last_block, ep1_first_block, blocks, validators, ep3_last_block=None, validators_n_1=None):
        self.ep2_last_block, self.ep1_first_block, self.ep3_last_block = ep2_last_block, ep1_first_block, ep3_last_block
        self.blocks, self.validators, self.validators_n_1 = list(blocks), list(validators), validators_n_1
        if len(self.blocks) not in (5, 6):
            raise ValueError("Invalid blocks.len() %d" % len(self.blocks))

    @classmethod
    def from_fixture(cls, win):
        """tests/golden/block_window_*.json (made by tests/golden/make_block_window_fixture.py from the reference's data/)"""
        hx = bytes.fromhex
        blocks = []
        for blk in win["blocks"]:
            f = {k: hx(blk[k]) for k in ("hash", "prev_hash", "epoch_id", "last_ds_final_hash", "last_final_hash")}
            f["height"] = blk["height"]
            f["approvals"] = [hx(a) for a in blk["approvals"]]
            blocks.append((f, hx(blk["bytes"])))
        pair = lambda k: (hx(win[k]["bytes"]), hx(win[k]["hash"])) if win.get(k) else None
        vn1 = [hx(v) for v in win["validators_n_1"]] if win.get("validators_n_1") else None
        return cls(pair("ep2_last_block"), pair("ep1_first_block"), blocks, [hx(v) for v in win["validators"]],
                   pair("ep3_last_block"), vn1)

    def bft_args(self):
        a = (self.ep2_last_block[0], self.ep2_last_block[1], self.ep1_first_block[0], self.ep1_first_block[1], self.blocks)
        kw = {}
        if self.ep3_last_block is not None:
            kw = {"ep3_last_block_bytes": self.ep3_last_block[0], "ep3_last_block_hash": self.ep3_last_block[1],
                  "validators_n_1": self.validators_n_1}
        return a, kw

    def approval_sets(self):
        """[(msg, approvals, validators)] in the order prove_block_bft reaches them (bft.rs:264-316: the finality proof of Block_i /
        Block_0 signed by the approvals in Block_i+1's successor header; :317-500: the one of Block_n-1)"""
        b = self.blocks
        sets = [(SG.generate_signed_message(b[4][0]["height"], b[3][0]["height"], b[3][0]["prev_hash"]), b[3][0]["approvals"],
                 self.validators)]
        if len(b) == 6:
            sets.append((SG.generate_signed_message(b[5][0]["height"], b[4][0]["height"], b[4][0]["prev_hash"]), b[4][0]["approvals"],
                         self.validators_n_1))
        return sets

    def expected_public_inputs(self):
        """public inputs of the final proof(s): [flag, hash(block), hash(ancestor), hash(ancestor)] (bft.rs:470-500)"""
        b = self.blocks
        if len(b) == 5:
            return [[0] + list(b[4][0]["hash"]) + list(self.ep2_last_block[1]) + list(self.ep1_first_block[1])]
        return [[1] + list(b[4][0]["hash"]) + list(self.ep2_last_block[1]) + list(self.ep1_first_block[1]),
                [1] + list(b[5][0]["hash"]) + list(self.ep3_last_block[1]) + list(self.ep2_last_block[1])]


def prewarm_jobs(windows, extra_msg_lens=()):
    """[(kind, message length)] of the cacheable circuits the block proofs of `windows` use: the Ed25519 circuit of every approval
    message length (prove_crypto/ed25519.rs:18-42) and the SHA-256 circuit (prove_crypto/sha256.rs:62-83) of every message the DAG
    hashes -- inner_lite (208 B), inner_rest of every header (block_finality.rs:98-198), the two 64-byte joins of a header-hash
    chain (header_bphash.rs:34-111), the borsh validator list (bp_hash, :121-139) and valid_keys (signatures.rs:107-112)."""
    from .prove_bft import INNER_LITE_BYTES, PK_HASH_BYTES, SIG_BYTES, TYPE_BYTE
    ed, sha = set(int(x) for x in extra_msg_lens), {INNER_LITE_BYTES, 2 * PK_HASH_BYTES}
    for w in ([windows] if isinstance(windows, BlockWindow) else list(windows)):
        vlists = [w.validators] + ([w.validators_n_1] if w.validators_n_1 else [])
        for validators in vlists:
            sha.add(4 + sum(len(v) for v in validators))
        for msg, approvals, _ in w.approval_sets():
            ed.add(len(msg))
            sha.add(33 * sum(1 for a in approvals if len(a) == 66))
        heads = [raw for _, raw in w.blocks] + [x[0] for x in (w.ep2_last_block, w.ep1_first_block, w.ep3_last_block) if x]
        for raw in heads:
            sha.add(len(raw) - (TYPE_BYTE + PK_HASH_BYTES + INNER_LITE_BYTES) - TYPE_BYTE - SIG_BYTES)
    return [("ed25519", n) for n in sorted(ed)] + [("sha256", n) for n in sorted(sha) if n > 0]


class _EdCircuit:
    """the reference's per-signature circuit for one message length (prove_crypto/ed25519.rs:18-42), resident once per prover stream,
    with its device witness interpreter and the double-buffered wire matrices"""
    pass


class _ApprovalStub:
    """what the DAG thread's BlockProver calls for `prove_approvals` / keys-stakes: the results of the pipeline's other threads"""

    def __init__(self, recursion):
        self.recursion = recursion
        self.sets = {}                   # message bytes -> _SetState of the block in its fold / DAG stage

    def keys_stakes_early(self, msg, approvals, validators):
        return self.sets[bytes(msg)].ks_future.result()

    def prove_approvals(self, msg, approvals, validators):
        from .plonky2 import HASH_GL
        from .plonky2 import serialization as S
        rc, raw, valid_keys = self.sets[bytes(msg)].future.result()
        return (rc, S.proof_from_bytes(raw, rc.common, HASH_GL)), valid_keys

    def close(self):
        self.recursion.close()


class _SetState:
    """one approval set of one block: filled by the signature stage, consumed by the fold thread"""

    def __init__(self, msg, approvals, validators):
        self.msg, self.approvals, self.validators = bytes(msg), approvals, validators
        self.future, self.ks_future = Future(), Future()
        self.local_agg = None


class BlockResult:
    """block / block_n_1: (common, verifier_only, proof json) triples of prove_block_bft; wrap / wrap_n_1: (RecursiveCircuit, proof
    bytes) of the BN128 wrap when asked for; timings in seconds"""

    def __init__(self):
        self.block = self.block_n_1 = self.wrap = self.wrap_n_1 = None
        self.t0 = self.t_signatures = self.t_done = None
        self.t_verify = self.witness_s = self.keys_stakes_s = 0.0
        self.fold_host_ms = {"inputs": 0.0, "witness": 0.0, "prove": 0.0}
        self.dag_seconds, self.dag_counts = {}, {}
        self.aggregates = []             # (RecursiveCircuit, raw closing proof, valid_keys) per approval set


class BlockPipeline:
    def __init__(self, device_id=0, prove_streams=4, witness_batch=32, host_witness=False, rank=0, world=1, comm_device=None,
                 host_threads=None, wrap=True, device_share=1):
        """One per process (= per GPU rank).  prove_streams: proofs in flight (prove_streams - 1 Ed25519 provers + the fold
        stream); witness_batch: signatures per device witness batch (<= 64; 0.49 GB of HBM each, two buffers per message length).
        rank / world / comm_device: the torch.distributed position for the strong form (collectives on `comm_device`).
        device_share: processes that share THIS device (functional multi-rank runs on a one-GPU box): the wire-matrix buffers are
        sized so that all of them fit (two buffers instead of three, the batch divided by the share)."""
        import zklc_amd
        from .keys_stakes import KeysStakesProver
        from .plonky2 import HASH_BN128, HASH_GL
        from .plonky2.recursion import RecursionProver
        from .prove_bft import BlockProver
        self.device_id, self.rank, self.world, self.comm_device = device_id, rank, world, comm_device
        self.nthreads = max(2, int(prove_streams))
        self.dev_wit = not host_witness
        self.wrap = wrap
        # wire-matrix buffers per circuit (each witness_batch x 0.49 GB of HBM).  Round 6: the kernel trace of the overlapped bench
        # showed all three prover queues idle TOGETHER for 0.3-0.4 s two to three times per block (profiles/r06h_*: the gaps of queues
        # 4-6 coincide) -- the provers had run out of witnesses: with two buffers the next block's first large batch can only be
        # produced once the previous block's last batch is fully proven.  Three buffers keep a batch ahead: 5.46 -> 5.18 s per block
        # on one box (profiles/r06k_*; within the noise on a second, r06l_*); giving the producer's stream the device's high priority on top measured WORSE (5.39-5.58 s)
        # and stays off.   ZKLC_WIT_BUFS / ZKLC_WIT_PRIORITY=1: A/B
        self.device_share = max(1, int(device_share))
        self.nbuf = max(2, int(os.environ.get("ZKLC_WIT_BUFS", "3" if self.device_share == 1 else "2")))
        if self.dev_wit:
            self.wchunk = max(1, min(64, int(witness_batch) // self.device_share))
        else:
            cores = host_threads or len(os.sched_getaffinity(0))
            self.wchunk = max(1, min(12 if world == 1 else 6, cores // max(1, world) - self.nthreads))
        self.ctx = zklc_amd.Context(device_id)                       # pre-check + the first Ed25519 prover
        self.ed_ctxs = [self.ctx] + [zklc_amd.Context(device_id) for _ in range(self.nthreads - 2)]
        self.wit_ctx = zklc_amd.Context(device_id, high_priority=os.environ.get("ZKLC_WIT_PRIORITY", "0") == "1") if self.dev_wit else None
        self.fold_ctx = zklc_amd.Context(device_id, high_priority=True)
        self.ks_ctx = zklc_amd.Context(device_id)
        self.dag_ctx = zklc_amd.Context(device_id, high_priority=True)
        self.rp = RecursionProver(self.fold_ctx, HASH_GL)
        self.ks_prover = KeysStakesProver(self.ks_ctx)
        self.stub = _ApprovalStub(RecursionProver(self.dag_ctx, HASH_GL))
        self.bprover = BlockProver(self.dag_ctx, self.stub)
        self.rpw = RecursionProver(self.dag_ctx, HASH_BN128) if wrap else None
        self.hdr_ctx = self.hprover = None        # the header thread of prove_stream (its own context and provers), on first use
        self._ed = {}
        self._lock = threading.Lock()
        self._sig_failed = False
        self.last = None

    # ------------------------------------------------------------------------------------------------ circuits
    def ed_circuit(self, msg_len):
        """build (or load from the circuit cache) the Ed25519 circuit of this message length and make it resident on every prover
        stream; two device wire-matrix buffers PER CIRCUIT (the device interpreter writes only the cells its program has slots for
        and relies on the rest staying zero, so two circuits never share a buffer)"""
        ent = self._ed.get(msg_len)
        if ent is not None:
            return ent
        import torch
        from .plonky2 import HASH_GL
        from .plonky2 import ed25519_circuit as E
        ent = _EdCircuit()
        ent.data, ent.targets, _ = E.build_cached(msg_len)
        ent.provers = [ent.data.prover(c, HASH_GL) for c in self.ed_ctxs]
        ent.common, ent.vd = ent.data.common_data(), ent.provers[0].verifier_data()
        nw, n_rows = ent.data.config["num_wires"], ent.data.n
        ent.free_slots = queue.Queue()
        for sl in range(self.nbuf):
            ent.free_slots.put(sl)
        ent.slot_left = [0] * self.nbuf
        if self.dev_wit:
            ent.dwit = ent.data.device_witness(self.wit_ctx)
            ent.d_bufs = [torch.zeros((self.wchunk, nw, n_rows), dtype=torch.int64, device="cuda:%d" % self.device_id)
                          for _ in range(self.nbuf)]
            torch.cuda.synchronize(self.device_id)       # the fill ran on torch's stream, the kernels run on the contexts'
        else:
            ent.pinned = [torch.zeros((self.wchunk, nw, n_rows), dtype=torch.int64).pin_memory() for _ in range(self.nbuf)]
            ent.views = [p.numpy().view(np.uint64) for p in ent.pinned]
        self._ed[msg_len] = ent
        return ent

    def prewarm(self, windows, extra_msg_lens=(), processes=None, timeout_s=900):
        """Cold start (round 5): the circuits a window needs that are NOT in the circuit cache yet (prewarm_jobs) are built by
        WORKER PROCESSES side by side (circuit construction is host Python: threads of one process share the GIL and build them
        one after the other, 180 s for the mainnet window; processes take the longest single build) and land in the cache, from
        which the pipeline then loads them.  A no-op without ZKLC_CIRCUIT_CACHE, when nothing is missing, or on a rank other
        than 0 (the other ranks wait on the entries' locks).  Never raises: what a worker did not build is built in this process
        on first use, as before.  Returns the report of circuit_cache.prewarm."""
        from .plonky2 import circuit_cache as CC
        if self.rank != 0:
            return {"skipped": "rank %d" % self.rank}
        return CC.prewarm(prewarm_jobs(windows, extra_msg_lens), processes=processes, timeout_s=timeout_s)

    # ------------------------------------------------------------------------------------------------ one block's state
    def _new_state(self, sets, strong):
        st = {"sets": [_SetState(*s) for s in sets], "errors": [], "ready": queue.Queue(), "res": BlockResult(), "strong": strong,
              "hdr_future": Future()}
        return st

    def _fail(self, st, e):
        st["errors"].append(e)
        self._sig_failed = True
        for s in st["sets"]:
            for fut in (s.future, s.ks_future):
                if not fut.done():
                    fut.set_exception(e)
            for ev in getattr(s, "ed_done", []):
                ev.set()
        if not st["hdr_future"].done():
            st["hdr_future"].set_exception(e)
        for _ in range(self.nthreads):
            st["ready"].put(None)

    def _precheck(self, st, ctx=None):
        """a3 for every approval set of the block + the witness inputs of the signatures this rank proves"""
        from .plonky2 import ed25519_circuit as E
        t0 = time.perf_counter()
        for s in st["sets"]:
            s.valid_keys, valid_pos, _, _ = SG.verify_approvals(ctx or self.ctx, s.msg, s.approvals, s.validators)   # raises InvalidSignature
            if not valid_pos:
                raise ValueError("no approvals present")
            _, pks, sigs = SG.slice_approvals(s.approvals, s.validators)
            s.n_sig = len(valid_pos)
            lo, hi = DIST.shard_range(s.n_sig, self.rank, self.world) if st["strong"] else (0, s.n_sig)
            s.my_sigs = list(range(lo, hi))
            s.ed = self.ed_circuit(len(s.msg))
            s.fills = {i: E.fill_ecdsa_targets(s.ed.targets, s.msg, sigs[i].tobytes(), pks[i].tobytes()) for i in s.my_sigs}
            if s.fills and s.ed.data._program is None:
                s.ed.data.witness_program(next(iter(s.fills.values())))
            s.ed_done = [threading.Event() for _ in range(s.n_sig)]
            s.ed_proofs = [None] * s.n_sig
        st["res"].t_verify = time.perf_counter() - t0

    # ------------------------------------------------------------------------------------------------ worker threads
    def _witness_producer(self, st):
        try:
            first = True
            for s in st["sets"]:
                ent, n_mine = s.ed, len(s.my_sigs)
                # a small first chunk (one signature per prover stream) so that proving starts after one witness time
                # (not for a block whose predecessor's proofs still occupy the provers: ZKLC_WIT_SMALL_FIRST=always restores it)
                small = first and (not st.get("has_prev") or os.environ.get("ZKLC_WIT_SMALL_FIRST") == "always")
                bounds = [0, min(n_mine, self.wchunk, max(1, self.nthreads - 1))] if small else [0]
                first = False
                while bounds[-1] < n_mine:
                    bounds.append(min(n_mine, bounds[-1] + self.wchunk))
                for c0, c1 in zip(bounds, bounds[1:]):
                    idx = s.my_sigs[c0:c1]
                    if not idx:
                        continue
                    sl = None
                    while sl is None:              # a failed block wakes the producer: no blocking get()
                        if st["errors"]:
                            return
                        try:
                            sl = ent.free_slots.get(timeout=0.05)
                        except queue.Empty:
                            pass
                    if st["errors"]:
                        ent.free_slots.put(sl)
                        return
                    t_ = time.perf_counter()
                    fills = [s.fills[i] for i in idx]
                    if self.dev_wit:
                        pis = ent.dwit.run(ent.d_bufs[sl].data_ptr(), fills, stream=self.wit_ctx.stream_ptr(), capacity=self.wchunk)
                    else:
                        assert len(idx) <= self.wchunk
                        _, pis = ent.data.generate_witness_native(fills, out=ent.views[sl][:len(idx)], threads=len(idx))
                    st["res"].witness_s += time.perf_counter() - t_
                    with self._lock:
                        ent.slot_left[sl] = len(idx)
                    for k, i in enumerate(idx):
                        st["ready"].put((s, i, sl, k, [int(x) for x in pis[k]]))
            for _ in range(self.nthreads):
                st["ready"].put(None)
        except Exception as e:
            self._fail(st, e)

    def _ed_worker(self, st, w):
        try:
            while True:
                item = st["ready"].get()
                if item is None:
                    return
                s, i, sl, k, pis = item
                ent = s.ed
                try:
                    if st["errors"]:
                        continue                   # drain: the block already failed, only the slot accounting matters
                    if self.dev_wit:
                        s.ed_proofs[i] = ent.provers[w].prove_dev(ent.d_bufs[sl][k].data_ptr(), pis, stream=self.ed_ctxs[w].stream_ptr())
                    else:
                        s.ed_proofs[i] = ent.provers[w].prove_host_ptr(ent.views[sl][k].ctypes.data, pis)
                    s.ed_done[i].set()
                finally:
                    with self._lock:
                        ent.slot_left[sl] -= 1
                        if ent.slot_left[sl] == 0:
                            ent.free_slots.put(sl)
        except Exception as e:
            self._fail(st, e)

    def _fold_worker(self, st):
        try:
            res = st["res"]
            for s in st["sets"]:
                agg = None
                for i in s.my_sigs:
                    s.ed_done[i].wait()
                    if st["errors"]:
                        return
                    nxt = (s.ed.common, s.ed.vd, s.ed_proofs[i])
                    if agg is None:
                        agg = nxt
                        continue
                    rc, proof = self.rp.recursive_proof(agg, nxt, raw=True)
                    for k in res.fold_host_ms:
                        res.fold_host_ms[k] += self.rp.last_host_ms[k]
                    agg = (rc.common, rc.verifier_only, proof)
                if st["strong"]:
                    s.local_agg = agg            # a (common, verifier_only, proof bytes) triple, or None without signatures
                    continue
                self._close_set(st, s, agg)
        except Exception as e:
            self._fail(st, e)

    def _close_set(self, st, s, agg):
        """signatures.rs:125-139: the closing proof carries sha256(valid_keys) as 32 public inputs"""
        rc, proof = self.rp.recursive_proof(agg, None, list(hashlib.sha256(s.valid_keys).digest()), raw=True)
        st["res"].t_signatures = time.perf_counter()
        st["res"].aggregates.append((rc, proof, s.valid_keys))
        s.future.set_result((rc, proof, s.valid_keys))

    def _ks_worker(self, st):
        try:
            t_ = time.perf_counter()
            for s in st["sets"]:
                s.ks_future.set_result(self.ks_prover.prove_valid_keys_stakes_in_validators_list(
                    s.valid_keys, hashlib.sha256(s.valid_keys).digest(), s.validators))
            st["res"].keys_stakes_s = time.perf_counter() - t_
        except Exception as e:
            self._fail(st, e)

    def _header_worker(self, st, jobs, owner):
        try:
            st["headers"] = {name: self.bprover.prove_header_job(jobs[name]) for name in jobs if owner[name] == self.rank}
        except Exception as e:
            self._fail(st, e)

    def _header_prover(self):
        """a second BlockProver on its own high-priority context, used for NOTHING but the block-header proofs (bft.rs:64-205) of the
        streaming form: they are independent leaves of the DAG, so the header thread of block b + 1 runs beside the joins and
        the wrap of block b instead of behind them"""
        if self.hprover is None:
            import zklc_amd
            from .plonky2 import HASH_GL
            from .plonky2.recursion import RecursionProver
            from .prove_bft import BlockProver
            self.hdr_ctx = zklc_amd.Context(self.device_id, high_priority=True)
            self.hprover = BlockProver(self.hdr_ctx, _ApprovalStub(RecursionProver(self.hdr_ctx, HASH_GL)))
        return self.hprover

    def _stream_header_worker(self, st, window):
        try:
            hp = self._header_prover()
            hp.counts, hp.seconds = {}, {}
            a, kw = window.bft_args()
            jobs = hp.header_jobs(*a, kw.get("ep3_last_block_bytes"), kw.get("ep3_last_block_hash"))
            got = {name: hp.prove_header_job(job) for name, job in jobs.items()}
            st["hdr_counts"], st["hdr_seconds"] = dict(hp.counts), dict(hp.seconds)
            st["hdr_future"].set_result(got)
        except Exception as e:
            self._fail(st, e)

    def _dag_worker(self, st, window, owner):
        try:
            res = st["res"]
            remote = None
            if st.get("hdr_thread"):   # streaming form: every header proof comes from the header thread
                remote = lambda name: st["hdr_future"].result()[name]
            elif st["strong"]:    # the header proofs of the other ranks arrive through hdr_future; rank 0's own are made here
                remote = lambda name: None if owner[name] == 0 else st["hdr_future"].result()[name]
            a, kw = window.bft_args()
            res.block, res.block_n_1 = self.bprover.prove_block_bft(*a, window.validators, header_proofs=remote, **kw)
            if self.rpw is not None:
                # bin/prove_block.rs:279-287: recursive_proof::<F, Cbn128, C, D>((..bi..), None, Some(&bi_proof.public_inputs))
                res.wrap = self.rpw.recursive_proof(res.block, None, list(res.block[2]["public_inputs"]), raw=True)
                if res.block_n_1 is not None:
                    res.wrap_n_1 = self.rpw.recursive_proof(res.block_n_1, None, list(res.block_n_1[2]["public_inputs"]), raw=True)
            res.dag_seconds, res.dag_counts = dict(self.bprover.seconds), dict(self.bprover.counts)
            for k, v in st.get("hdr_counts", {}).items():        # what the header thread proved belongs to the block's DAG as well
                res.dag_counts[k] = res.dag_counts.get(k, 0) + v
            for k, v in st.get("hdr_seconds", {}).items():
                res.dag_seconds[k] = res.dag_seconds.get(k, 0.0) + v
            res.t_done = time.perf_counter()
        except Exception as e:
            self._fail(st, e)

    # ------------------------------------------------------------------------------------------------ stages
    @staticmethod
    def _thread_name(f, a):
        # worker threads carry their role in their name (diagnostics: bench.py's ZKLC_BENCH_SAMPLE reads it)
        if getattr(f, "__name__", "") == "_after" and len(a) == 3:
            f, a = a[1], a[2]
        n = getattr(f, "__name__", "worker").lstrip("_")
        return "zklc-%s%s" % (n, "-%d" % a[1] if n == "ed_worker" and len(a) > 1 else "")

    def _start(self, fns):
        ths = [threading.Thread(target=f, args=a, name=self._thread_name(f, a)) for f, a in fns]
        for th in ths:
            th.start()
        return ths

    def _reset_slots(self, ent):
        """every wire-matrix buffer free again (a block that failed mid-way may have left a slot half-consumed)"""
        with self._lock:
            while True:
                try:
                    ent.free_slots.get_nowait()
                except queue.Empty:
                    break
            for sl in range(self.nbuf):
                ent.slot_left[sl] = 0
                ent.free_slots.put(sl)

    def _start_signature_stage(self, st):
        st["res"].t0 = time.perf_counter()
        self._precheck(st)
        if self._sig_failed:                     # the previous signature stage ended in an error: its slot accounting is void
            for ent in self._ed.values():
                self._reset_slots(ent)
            self._sig_failed = False
        return self._start([(self._witness_producer, (st,))] + [(self._ed_worker, (st, w)) for w in range(len(self.ed_ctxs))])

    def _start_signature_stage_after(self, st, prev):
        """streaming form: the producer and the prover of stream w start when THEIR predecessors of the previous block (`prev`: the
        list _start_signature_stage returned for it) have ended -- the next block's first witnesses are made while this block's last
        batch is still being proven, so the prover streams go from one block to the next without waiting for a witness batch"""
        st["res"].t0 = time.perf_counter()
        # the batched pre-check runs on the witness producer's context: the producer of the previous block has ended (the caller
        # waited for it) while the first prover stream -- self.ctx -- may still be in the middle of a proof
        self._precheck(st, self.wit_ctx)
        st["has_prev"] = prev is not None
        prev = prev or [None] * (1 + len(self.ed_ctxs))
        return self._start([(self._after, ([prev[0]] if prev[0] else [], self._witness_producer, (st,)))] +
                           [(self._after, ([prev[1 + w]] if prev[1 + w] else [], self._ed_worker, (st, w))) for w in range(len(self.ed_ctxs))])

    def _begin_dag_stage(self, st):
        """the fold / DAG / keys-stakes stage of a block owns the stub's futures and the block prover's counters"""
        self.stub.sets = {s.msg: s for s in st["sets"]}
        self.bprover.counts, self.bprover.seconds = {}, {}

    def _start_dag_stage(self, st, window):
        self._begin_dag_stage(st)
        return self._start([(self._fold_worker, (st,)), (self._dag_worker, (st, window, {})), (self._ks_worker, (st,))])

    @staticmethod
    def _join(ths):
        for th in ths:
            th.join()

    @staticmethod
    def _raise(st):
        if st["errors"]:
            raise st["errors"][0]

    def _strong_checkpoint(self, st, pending):
        """strong form: AND of the ranks' success flags before an exchange.  If any rank failed, every rank fails its block (the
        futures the DAG / fold threads wait on get the exception), joins its threads and raises -- the failing rank its own error,
        the others RemoteRankFailed."""
        if DIST.all_ok(not st["errors"], device=self.comm_device):
            return
        if not st["errors"]:
            self._fail(st, DIST.RemoteRankFailed("rank %d: another rank failed its part of the block" % self.rank))
        self._join(pending)
        self._raise(st)

    # ------------------------------------------------------------------------------------------------ public API
    def prove_block_bft(self, window, strong=False):
        """ONE block, every stage concurrently -> BlockResult (rank 0 in the strong form; the other ranks return None)."""
        strong = bool(strong) and self.world > 1
        st = self._new_state(window.approval_sets(), strong)
        a, kw = window.bft_args()
        jobs = self.bprover.header_jobs(*a, kw.get("ep3_last_block_bytes"), kw.get("ep3_last_block_hash"))
        owner = DIST.assign_jobs(list(jobs), self.world) if strong else {}
        ks_rank = self.world - 1 if strong else self.rank
        self._begin_dag_stage(st)
        sig = self._start_signature_stage(st) + self._start([(self._fold_worker, (st,))])
        dag = self._start([(self._dag_worker, (st, window, owner))]) if (not strong or self.rank == 0) else []
        side = self._start([(self._ks_worker, (st,))]) if self.rank == ks_rank else []
        if strong and self.rank != 0:
            side += self._start([(self._header_worker, (st, jobs, owner))])
        if strong:
            # (1) the header proofs and the keys / stakes proofs of the other ranks travel to rank 0 (point-to-point, ~150 KB each)
            self._join(side)
            self._strong_checkpoint(st, sig + dag)    # a rank whose header / keys-stakes job failed fails the block on EVERY rank
            mine = {"headers": st.get("headers", {}),
                    "ks": [s.ks_future.result() for s in st["sets"]] if self.rank == ks_rank else None}
            parts = DIST.gather_objects(mine if self.rank != 0 else None, 0, device=self.comm_device)
            if self.rank == 0:
                merged = {}
                for part in parts[1:]:
                    merged.update(part["headers"])
                    if part["ks"] is not None and ks_rank != 0:
                        for s, ks in zip(st["sets"], part["ks"]):
                            s.ks_future.set_result(ks)
                st["hdr_future"].set_result(merged)
            # (2) local folds -> binary tree over the ranks -> closing proof on rank 0, per approval set
            self._join(sig)
            self._strong_checkpoint(st, dag)          # .. and so does a failed signature stage: no partial aggregate is folded
            fold_err = []

            def combine(x, y):
                # a combine that raised would leave this rank's tree partners blocked in recv_obj: the exchange pattern is kept
                # (every send / receive still happens, with nothing folded), the error is recorded and checkpoint (3) below makes
                # EVERY rank fail the block together
                if fold_err:
                    return None
                try:
                    rc_, p_ = self.rp.recursive_proof(x, y, raw=True)
                    return (rc_.common, rc_.verifier_only, p_)
                except Exception as e:
                    fold_err.append(e)
                    return None
            totals = [DIST.tree_fold(s.local_agg, combine, device=self.comm_device) for s in st["sets"]]   # None = no signatures here
            if fold_err:
                self._fail(st, fold_err[0])
            self._strong_checkpoint(st, dag)          # (3) the tree fold succeeded on every rank
            if self.rank == 0:
                try:
                    for s, total in zip(st["sets"], totals):
                        if not s.future.done():       # rank 0's DAG thread may have failed meanwhile: _fail has set the futures
                            self._close_set(st, s, total)
                except Exception as e:
                    self._fail(st, e)
            self._join(dag)
            self._strong_checkpoint(st, [])           # (4) closing proofs, joins and wrap on rank 0: the block succeeds or fails everywhere
        else:
            self._join(sig + dag + side)
        self._raise(st)
        self.last = st["res"]
        return st["res"] if (not strong or self.rank == 0) else None

    def prove_stream(self, windows, on_block_done=None):
        """Consecutive blocks as a software pipeline (a light client proves a stream of blocks).  Every worker of block b -- the
        witness producer, the prover of stream w, the fold thread, the header thread, the keys / stakes thread, the DAG thread (joins
        and the BN128 wrap) -- owns its context and its provers and is chained to ITS predecessor of block b - 1 only:
          * the fold thread and the keys / stakes thread start with the signature stage; the fold consumes the Ed25519 proofs as
            they arrive instead of working through a backlog;
          * the block-header proofs (independent leaves of the DAG, ~4 s of small proofs) have a thread and provers of their own
            (`_header_prover`): the headers of block b + 1 are proven beside the joins and the wrap of block b, not behind them;
          * the loop goes on to block b + 1 when the witness PRODUCER of block b has ended, i.e. while its last batch is still being
            proven: the next block's first witnesses are ready before the prover streams run dry.
        Rounds 3-5 chained whole stages (the fold / DAG stage of a block after the previous block's, the next signature stage after
        the last signature proof): once the faster leaf hash of round 6 had made the serial chains the longer path, the GPU idled
        between blocks (87-95 % busy, block latencies of 9-12 s: profiles/r06c_*, r06e_*).
        Returns the BlockResults in order; `on_block_done(result)` is called as each block completes.  A block that fails ends the
        stream: the blocks before it are delivered, its error is raised, the block started after it is abandoned.

        Host policy for the duration of the stream (round 6, profiles/r06r_gc_ab.txt; both restored on return): the cyclic
        collector's automatic runs stop EVERY thread of the pipeline -- measured 0.25-0.29 s of pauses per block, five full
        collections of ~40 ms among them, which are the > 20 ms holes of the kernel trace -- so they are replaced by ONE
        young-generation collection per delivered block (ZKLC_STREAM_GC=auto keeps the interpreter's thresholds: 5.83 s per block
        against 5.2-5.5 on one box, the GPU 86 % instead of 91-96 % busy);
        the interpreter's thread switch interval is 0.5 ms instead of 5 ms (ZKLC_SWITCH_INTERVAL_MS; 0 = leave it): a worker that
        returns from a proof waits up to one interval for the GIL while another thread runs Python -- a 2 ms sleeper overslept
        0.34 s per block in total at 5 ms and 0.07 s at 0.5 ms; the block time moved inside the run-to-run noise (mid-run mean
        5.31 -> 5.17 s, profiles/r06s_switch_interval_ab.txt).
        A host that streams without end should move its long-lived objects out of the collector's sight once after start-up
        (gc.freeze(), as bench.py does: the circuits are tens of millions of objects) and run a full collection when IT chooses:
        objects that survive the per-block collection are only looked at again by a full one."""
        import gc
        mode, sw = os.environ.get("ZKLC_STREAM_GC", "block"), os.environ.get("ZKLC_SWITCH_INTERVAL_MS", "0.5")
        sw = sw if float(sw) > 0 else None
        manual = mode == "block" and gc.isenabled()
        old_sw = sys.getswitchinterval()
        cb = on_block_done
        if manual:
            gc.disable()

            def cb(res):
                if on_block_done:
                    on_block_done(res)
                gc.collect(1)
        if sw:
            sys.setswitchinterval(float(sw) / 1e3)
        try:
            return self._prove_stream(windows, cb)
        finally:
            if manual:
                gc.enable()
            if sw:
                sys.setswitchinterval(old_sw)

    def _prove_stream(self, windows, on_block_done):
        results, prev_dag, prev_st = [], [], None
        prev_fold, prev_ks, prev_hdr, prev_sig, all_threads = [], [], [], None, []
        # A/B (ZKLC_STREAM, a comma list; default "fold,hdr,ahead"; "stages" = none of them = rounds 3-5): fold = the fold and keys /
        # stakes threads start with the signature stage; hdr = the header thread; ahead = the producer lookahead
        opts = set(x for x in os.environ.get("ZKLC_STREAM", "fold,hdr,ahead").split(",") if x and x != "stages")
        o_fold, o_hdr, o_ahead = "fold" in opts, "hdr" in opts, "ahead" in opts

        def deliver(st_):
            results.append(st_["res"])
            if on_block_done:
                on_block_done(st_["res"])
        for window in windows:
            st = self._new_state(window.approval_sets(), False)
            st["hdr_thread"] = o_hdr
            try:
                if self._sig_failed and o_ahead:  # a failed signature stage leaves the slot accounting void: start from a clean state
                    self._join(all_threads)
                    for ent in self._ed.values():
                        self._reset_slots(ent)
                    self._sig_failed, prev_sig = False, None
                sig = self._start_signature_stage_after(st, prev_sig) if o_ahead else self._start_signature_stage(st)
            except Exception:
                self._join(all_threads)
                raise
            early = []
            if o_fold:
                early += [(prev_fold, self._fold_worker, (st,)), (prev_ks, self._ks_worker, (st,))]
            if o_hdr:
                early += [(prev_hdr, self._stream_header_worker, (st, window))]
            started = [self._start([(self._after, e)])[0] for e in early]
            all_threads = [t for t in all_threads if t.is_alive()] + sig + started
            self._join(prev_dag)                  # the DAG thread of block b - 1 (the stub's futures, the block prover's counters)
            if prev_st is not None:
                if prev_st["errors"]:             # block b - 1 failed: abandon block b, raise b - 1's error
                    self._fail(st, RuntimeError("the previous block of the stream failed"))
                    self._join(all_threads)
                    self._raise(prev_st)
                deliver(prev_st)
            self._begin_dag_stage(st)
            late = [(self._dag_worker, (st, window, {}))]
            if not o_fold:
                late += [(self._fold_worker, (st,)), (self._ks_worker, (st,))]
            dag = self._start(late)
            all_threads += dag
            prev_dag = dag + started
            if o_fold:
                prev_fold, prev_ks = [started[0]], [started[1]]
            if o_hdr:
                prev_hdr = [started[-1]]
            prev_sig, prev_st = sig, st
            if o_ahead:
                sig[0].join()                     # the witness producer of this block
            else:
                self._join(sig)
        self._join(all_threads)
        if prev_st is not None:
            self._raise(prev_st)
            deliver(prev_st)
            self.last = prev_st["res"]
        return results

    def _after(self, earlier, fn, args):
        """run fn(*args) once the threads `earlier` (the same worker of the previous block) have ended"""
        self._join(earlier)
        fn(*args)

    def prove_approvals(self, msg, approvals, validators):
        """`prove_approvals` (signatures.rs:43-141) alone through the pipeline's signature stage and fold thread:
        -> ((RecursiveCircuit, closing proof bytes), valid_keys).  BASELINE configs[4] (a synthetic epoch of N validators) runs
        through this: the keys / stakes circuit cannot express N > 255 positions, the signature aggregate can."""
        st = self._new_state([(msg, approvals, validators)], False)
        ths = self._start_signature_stage(st) + self._start([(self._fold_worker, (st,))])
        self._join(ths)
        self._raise(st)
        rc, raw, valid_keys = st["sets"][0].future.result()
        self.last = st["res"]
        return (rc, raw), valid_keys

    def close(self):
        for ent in self._ed.values():
            for pr in ent.provers:
                pr.close()
            if self.dev_wit:
                ent.dwit.close()
                ent.d_bufs = None
        self._ed = {}
        self.rp.close()
        if self.rpw is not None:
            self.rpw.close()
        self.bprover.close()              # closes the stub's recursion prover, the SHA-256 and primitive provers of the DAG thread
        if self.hprover is not None:
            self.hprover.close()
            self.hdr_ctx.close()
            self.hprover = self.hdr_ctx = None
        self.ks_prover.close()
        for c in self.ed_ctxs[1:] + [x for x in (self.wit_ctx, self.fold_ctx, self.ks_ctx, self.dag_ctx) if x is not None]:
            c.close()
        self.ctx.close()
