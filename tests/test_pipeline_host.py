"""Host-side pieces of zklc_amd/pipeline.py that need no GPU: `BlockWindow` = the arguments of `prove_block_bft`
(near_bft_finality/src/prove_bft/bft.rs:38-62) read from the mainnet fixtures, its approval sets (bft.rs:264-316, :317-500) and the
public inputs of the final proofs (bft.rs:470-500)."""
import glob
import json
import os

import pytest

from zklc_amd import signatures as SG
from zklc_amd.pipeline import BlockWindow

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "block_window_*.json")))


@pytest.mark.parametrize("path", FIXTURES, ids=os.path.basename)
def test_block_window_from_the_fixtures(path):
    win = BlockWindow.from_fixture(json.load(open(path)))
    n = len(win.blocks)
    assert n in (5, 6)
    args, kw = win.bft_args()
    assert len(args) == 5 and args[4] is win.blocks and (n == 6) == bool(kw)
    sets = win.approval_sets()
    assert len(sets) == n - 4
    b = win.blocks
    # the approvals of a block's successor header sign (parent hash, parent height, target height) of the block being finalised
    msg, approvals, validators = sets[0]
    assert msg == SG.generate_signed_message(b[4][0]["height"], b[3][0]["height"], b[3][0]["prev_hash"])
    assert len(msg) == 41 and msg[0] == 0 and msg[1:33] == b[3][0]["prev_hash"]
    assert approvals is b[3][0]["approvals"] and validators is win.validators and len(approvals) == len(validators)
    if n == 6:
        msg1, approvals1, validators1 = sets[1]
        assert msg1[1:33] == b[4][0]["prev_hash"] and approvals1 is b[4][0]["approvals"] and validators1 is win.validators_n_1
    pis = win.expected_public_inputs()
    assert len(pis) == n - 4 and all(len(p) == 97 and p[0] == (1 if n == 6 else 0) for p in pis)
    assert bytes(pis[0][1:33]) == b[4][0]["hash"] and bytes(pis[0][33:65]) == win.ep2_last_block[1]
    # the hash of a block is the hash of its borsh bytes' pieces: the fixture's fields are consistent with its bytes
    from zklc_amd import header_bphash as H
    if hasattr(H, "block_hash_from_bytes"):
        assert H.block_hash_from_bytes(b[4][1]) == b[4][0]["hash"]


def test_block_window_rejects_other_lengths():
    win = BlockWindow.from_fixture(json.load(open(FIXTURES[0])))
    with pytest.raises(ValueError, match="Invalid blocks.len"):
        BlockWindow(win.ep2_last_block, win.ep1_first_block, win.blocks[:4], win.validators)


def test_prewarm_jobs_of_the_mainnet_window():
    """pipeline.prewarm_jobs: the cacheable circuits of a window -- the approval message's Ed25519 circuit and one SHA-256 circuit
    per hashed message length (inner_lite, every inner_rest, the 64-byte joins, the borsh validator list, valid_keys)"""
    import hashlib
    from zklc_amd.pipeline import BlockWindow, prewarm_jobs
    from zklc_amd.plonky2.sha256 import block_num_of
    load_golden = lambda name: json.load(open(os.path.join(GOLDEN, name)))
    win = BlockWindow.from_fixture(load_golden("block_window_HPi5.json"))
    jobs = prewarm_jobs(win, extra_msg_lens=[17])
    assert [j for j in jobs if j[0] == "ed25519"] == [("ed25519", 17), ("ed25519", 41)]
    sha = [n for k, n in jobs if k == "sha256"]
    assert 208 in sha and 64 in sha and 33 * 73 in sha
    assert 4 + sum(len(v) for v in win.validators) in sha
    # every header's inner_rest: hash(header) = sha256(sha256(sha256(inner_lite) || sha256(inner_rest)) || prev_hash) pins the slicing
    for f, raw in win.blocks:
        lite, rest = raw[33:33 + 208], raw[33 + 208:len(raw) - 65]
        assert len(rest) in sha
        inner = hashlib.sha256(hashlib.sha256(lite).digest() + hashlib.sha256(rest).digest()).digest()
        assert hashlib.sha256(inner + raw[1:33]).digest() == f["hash"]
    assert len({block_num_of(n) for n in sha}) <= len(sha)
    epoch = BlockWindow.from_fixture(load_golden("block_window_epoch_CRTZ.json"))
    assert len([j for j in prewarm_jobs(epoch) if j[0] == "sha256"]) >= 6


# ---- the signature stage's thread protocol on stand-ins (no GPU): ADVICE r04 -- a failing prover must not leak its wire-matrix slot
# or leave the witness producer blocked, and the first witness chunk must fit the buffers
def _fake_signature_stage(n_sig, nthreads, wchunk, fail_on=None):
    import queue
    import threading
    import numpy as np
    from zklc_amd import pipeline as PL
    p = object.__new__(PL.BlockPipeline)
    p.nthreads, p.wchunk, p.nbuf, p.dev_wit = nthreads, wchunk, 2, False
    p._lock, p._sig_failed = threading.Lock(), False
    p.ed_ctxs = [None] * (nthreads - 1)
    chunks, proved = [], []

    class Data:
        def generate_witness_native(self, fills, out, threads):
            assert len(fills) <= len(out) <= wchunk, "a witness chunk larger than the buffer"
            chunks.append(len(fills))
            return None, [[7]] * len(fills)

    class Prover:
        def prove_host_ptr(self, ptr, pis):
            proved.append(ptr)
            if fail_on is not None and len(proved) > fail_on:
                raise RuntimeError("stand-in prover failure")
            return b"proof"
    ent = PL._EdCircuit()
    ent.data, ent.provers = Data(), [Prover() for _ in p.ed_ctxs]
    ent.views = [np.zeros((wchunk, 1, 1), dtype=np.uint64) for _ in range(p.nbuf)]
    ent.free_slots, ent.slot_left = queue.Queue(), [0] * p.nbuf
    for sl in range(p.nbuf):
        ent.free_slots.put(sl)
    p._ed = {41: ent}
    st = p._new_state([(b"m" * 41, [], [])], False)
    s = st["sets"][0]
    s.ed, s.n_sig, s.my_sigs = ent, n_sig, list(range(n_sig))
    s.fills = {i: {} for i in range(n_sig)}
    s.ed_done = [threading.Event() for _ in range(n_sig)]
    s.ed_proofs = [None] * n_sig
    ths = p._start([(p._witness_producer, (st,))] + [(p._ed_worker, (st, w)) for w in range(len(p.ed_ctxs))])
    for th in ths:
        th.join(30)
    assert not any(th.is_alive() for th in ths), "the signature stage did not terminate"
    return p, ent, st, s, chunks


def test_signature_stage_first_chunk_fits_the_buffer_and_all_proofs_are_made():
    p, ent, st, s, chunks = _fake_signature_stage(n_sig=11, nthreads=8, wchunk=2)          # prove_streams - 1 = 7 > witness_batch = 2
    assert not st["errors"] and max(chunks) <= 2 and sum(chunks) == 11
    assert all(x == b"proof" for x in s.ed_proofs) and all(ev.is_set() for ev in s.ed_done)
    assert ent.slot_left == [0, 0] and ent.free_slots.qsize() == 2


def test_signature_stage_with_failing_provers_terminates_and_frees_its_slots():
    p, ent, st, s, chunks = _fake_signature_stage(n_sig=9, nthreads=3, wchunk=2, fail_on=1)   # every prover fails after the first proof
    assert st["errors"] and "stand-in prover failure" in str(st["errors"][0])
    assert p._sig_failed
    p._reset_slots(ent)                      # what the next signature stage does first
    assert ent.slot_left == [0, 0] and sorted(ent.free_slots.get_nowait() for _ in range(2)) == [0, 1]
    import pytest as _pytest
    with _pytest.raises(RuntimeError):
        p._raise(st)


# ---- prove_stream as a software pipeline (round 6) on stand-in workers: who may run beside whom, in which order results come out,
# and what a failing block does to the stream
def _fake_stream(n_blocks, fail_block=None, opts=None, monkeypatch=None):
    import threading
    import time
    from zklc_amd import pipeline as PL
    if opts is not None:
        monkeypatch.setenv("ZKLC_STREAM", opts)
    p = object.__new__(PL.BlockPipeline)
    p.ed_ctxs, p._sig_failed, p._ed, p.last, p.wit_ctx = [None, None], False, {}, None, None
    p.stub = type("S", (), {})()
    p.bprover = type("B", (), {"counts": {}, "seconds": {}})()
    log, lock, running = [], threading.Lock(), {}

    def work(kind, b, seconds, st):
        with lock:
            running[kind] = running.get(kind, 0) + 1
            assert running[kind] == 1, "two %s workers at once: they share provers" % kind
            log.append(("start", kind, b))
        time.sleep(seconds)
        with lock:
            running[kind] -= 1
            log.append(("end", kind, b))
    blocks = iter(range(n_blocks))
    p._new_state = lambda sets, strong: {"sets": [], "errors": [], "res": PL.BlockResult(), "b": next(blocks), "strong": False,
                                         "hdr_future": PL.Future(), "ready": __import__("queue").Queue()}
    p._precheck = lambda st, ctx=None: None
    p._witness_producer = lambda st: work("producer", st["b"], 0.03, st)
    p._ed_worker = lambda st, w: work("ed%d" % w, st["b"], 0.06, st)
    p._ks_worker = lambda st: work("ks", st["b"], 0.01, st)
    p._stream_header_worker = lambda st, window: work("hdr", st["b"], 0.05, st)
    p._begin_dag_stage = lambda st: None

    def fold(st):
        work("fold", st["b"], 0.07, st)
        if st["b"] == fail_block:
            p._fail(st, RuntimeError("stand-in fold failure in block %d" % st["b"]))
    p._fold_worker = fold

    def dag(st, window, owner):
        while not any(e == ("end", "fold", st["b"]) for e in list(log)) and not st["errors"]:
            time.sleep(0.002)                 # the DAG thread waits for the block's aggregate
        work("dag", st["b"], 0.03, st)
        st["res"].t_done = time.perf_counter()
    p._dag_worker = dag
    p.nthreads = 3
    done = []
    win = type("W", (), {"approval_sets": lambda self: []})()
    err = None
    try:
        res = p.prove_stream([win] * n_blocks, lambda r: done.append(r))
    except RuntimeError as e:
        res, err = None, e
    assert threading.active_count() <= 2, "prove_stream left worker threads behind"
    return p, log, res, done, err


def test_prove_stream_chains_every_worker_to_its_own_predecessor(monkeypatch):
    p, log, res, done, err = _fake_stream(4, monkeypatch=monkeypatch)
    assert err is None and len(res) == 4 and done == res
    pos = {e: i for i, e in enumerate(log)}
    for kind in ("producer", "ed0", "ed1", "fold", "ks", "hdr", "dag"):
        ends = [pos[("end", kind, b)] for b in range(4)]
        starts = [pos[("start", kind, b)] for b in range(4)]
        assert all(starts[b + 1] > ends[b] for b in range(3)), "%s of block b + 1 started before block b's ended" % kind
    # the point of the software pipeline: the NEXT block's producer, provers, fold and headers run beside THIS block's DAG thread
    for b in range(3):
        assert pos[("start", "producer", b + 1)] < pos[("end", "dag", b)]
        assert pos[("start", "fold", b + 1)] < pos[("end", "dag", b)]
        assert pos[("start", "hdr", b + 1)] < pos[("end", "dag", b)]
    # .. and the lookahead: the producer of block b + 1 starts while a prover stream of block b is still busy
    assert any(pos[("start", "producer", b + 1)] < pos[("end", "ed0", b)] for b in range(3))
    assert p.last is res[-1]


def test_prove_stream_stage_chained_form_is_still_selectable(monkeypatch):
    p, log, res, done, err = _fake_stream(3, opts="stages", monkeypatch=monkeypatch)
    assert err is None and len(res) == 3
    pos = {e: i for i, e in enumerate(log)}
    assert not any(e[1] == "hdr" for e in log)                                    # no header thread
    for b in range(2):                                                            # rounds 3-5: stage after stage
        assert pos[("start", "fold", b + 1)] > pos[("end", "dag", b)]
        assert pos[("start", "producer", b + 1)] > max(pos[("end", "ed0", b)], pos[("end", "ed1", b)])


def test_prove_stream_delivers_the_blocks_before_a_failed_one_and_raises(monkeypatch):
    p, log, res, done, err = _fake_stream(4, fail_block=1, monkeypatch=monkeypatch)
    assert err is not None and "block 1" in str(err)
    assert len(done) == 1                                                         # block 0 was delivered, nothing after it
    assert not any(e[2] == 3 for e in log), "the stream went on after the failed block"


def test_prove_stream_host_policy_is_scoped_to_the_stream(monkeypatch):
    """for the duration of a stream the automatic collections are replaced by one young-generation collection per delivered block
    and the switch interval is what ZKLC_SWITCH_INTERVAL_MS says; both are the caller's again afterwards -- also after a failed
    stream -- and ZKLC_STREAM_GC=auto leaves the collector alone"""
    import gc
    import sys
    seen = []
    real_collect = gc.collect
    monkeypatch.setattr(gc, "collect", lambda *a: (seen.append((a, gc.isenabled(), sys.getswitchinterval())), real_collect(*a))[1])
    monkeypatch.setenv("ZKLC_SWITCH_INTERVAL_MS", "1")
    sw0 = sys.getswitchinterval()
    assert gc.isenabled()
    p, log, res, done, err = _fake_stream(3, monkeypatch=monkeypatch)
    assert err is None and len(done) == 3
    assert [s for s in seen if s[0] == (1,)] == [((1,), False, 0.001)] * 3       # one per block, collector off, interval set
    assert gc.isenabled() and sys.getswitchinterval() == sw0
    seen.clear()
    p, log, res, done, err = _fake_stream(3, fail_block=1, monkeypatch=monkeypatch)
    assert err is not None and gc.isenabled() and sys.getswitchinterval() == sw0
    seen.clear()
    monkeypatch.setenv("ZKLC_STREAM_GC", "auto")
    monkeypatch.setenv("ZKLC_SWITCH_INTERVAL_MS", "0")
    p, log, res, done, err = _fake_stream(2, monkeypatch=monkeypatch)
    assert err is None and not [s for s in seen if s[0] == (1,)] and gc.isenabled() and sys.getswitchinterval() == sw0
