"""world_size-2 gloo test of the multi-GPU sharding logic (CPU; the oracle stands in for the kernels)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cport
    import importlib
    D = importlib.import_module("zk-light-client-implementation_amd.distributed")
    pts = cport.bn254_gen_points(n, 3, 5)
    rng = np.random.default_rng(7)
    sc = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] >>= np.uint64(4)
    lo, hi = D.shard_range(n, rank, world)
    f = lambda p, s: cport.bn254_msm(p, s)[:2]
    out, inf = D.msm_sharded(f, f, pts[lo:hi], sc[lo:hi])
    # signatures shard: every rank verifies its own slice, the bitmap is the concatenation
    from oracle import ed25519_ref as ref
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from edcases import synthetic_set
    pks, sigs, msg = synthetic_set(10, seed=3, corrupt_every=4)
    a, b = D.shard_range(10, rank, world)
    mine = torch.tensor([int(cport.ed25519_verify(pks[i], sigs[i], msg)) for i in range(a, b)], dtype=torch.int64)
    sizes = [D.shard_range(10, r, world)[1] - D.shard_range(10, r, world)[0] for r in range(world)]
    parts = [torch.zeros(s, dtype=torch.int64) for s in sizes]
    dist.all_gather(parts, mine) if len(set(sizes)) == 1 else None
    q.put((rank, out.tolist(), inf, torch.cat(parts).tolist() if len(set(sizes)) == 1 else None))
    dist.destroy_process_group()


def test_shard_range():
    import importlib
    D = importlib.import_module("zk-light-client-implementation_amd.distributed")
    for n in [0, 1, 7, 8, 100, 819200]:
        for world in [1, 2, 3, 8]:
            r = [D.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_msm_sharded_world2_matches_single():
    sys.path.insert(0, ROOT)
    from oracle import cport
    n, world = 501, 2
    pts = cport.bn254_gen_points(n, 3, 5)
    rng = np.random.default_rng(7)
    sc = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] >>= np.uint64(4)
    want, winf, _ = cport.bn254_msm(pts, sc)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, inf, bitmap in res:
        assert inf == winf and out == want.tolist()
        assert bitmap == [int(i % 4 != 3) for i in range(10)]


def _proof_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    D = importlib.import_module("zk-light-client-implementation_amd.distributed")
    calls = []

    def prove_one(i):          # stands in for one Ed25519-circuit proof: ragged lengths, content tied to the index
        calls.append(i)
        return bytes([(7 * i + k) & 0xFF for k in range(100 + 13 * (i % 5))])
    out = D.prove_signatures_sharded(prove_one, n)
    q.put((rank, calls, [p.hex() for p in out], D.gather_bytes([]) == [[] for _ in range(world)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 2, 1])
def test_signature_proofs_sharded_over_two_ranks(n):
    """signature i -> rank i mod world, all-gather of the (ragged) proof bytes, every rank ends with the n proofs in
    signature order -- what the left fold consumes"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_proof_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
    want = [bytes([(7 * i + k) & 0xFF for k in range(100 + 13 * (i % 5))]).hex() for i in range(n)]
    for rank, calls, out, empty_ok in res:
        assert calls == list(range(rank, n, world)) and out == want and empty_ok


def _tree_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    D = importlib.import_module("zk-light-client-implementation_amd.distributed")
    lo, hi = D.shard_range(n, rank, world)
    # stand-in for this rank's left fold over its contiguous run of signature proofs: the structure of the aggregation tree
    local = None
    for i in range(lo, hi):
        local = ("leaf", i, bytes([i]) * (1000 + i)) if local is None else ("fold", local, ("leaf", i, bytes([i]) * (1000 + i)))
    combines = []

    def combine(a, b):
        combines.append(1)
        return ("fold", a, b)
    total = D.tree_fold(local, combine)
    jobs = D.assign_jobs(["ep2_lb", "ep1_fb", "b4", "b3", "b2", "b1", "bi0"], world)
    mine = {name: ("header", name, rank) for name, r in jobs.items() if r == rank}
    gathered = D.gather_objects(mine, 0)
    q.put((rank, total, len(combines), gathered, jobs))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 7), (3, 7), (4, 3), (2, 1)])
def test_tree_fold_and_header_gather(world, n):
    """contiguous signature shards folded locally, log2(world) point-to-point exchanges of one aggregate each,
    the leaves of the resulting tree in signature order on rank 0; ranks without signatures pass None up the tree; the
    header proofs made by the other ranks arrive on rank 0"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_tree_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda x: x[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0

    def leaves(t):
        return [t[1]] if t[0] == "leaf" else leaves(t[1]) + leaves(t[2])
    rank0 = res[0]
    assert leaves(rank0[1]) == list(range(n))
    assert all(r[1] is None for r in res[1:])
    # number of recursive_proof calls in the tree phase = (ranks that hold signatures) - 1
    import importlib
    D = importlib.import_module("zk-light-client-implementation_amd.distributed")
    holders = sum(1 for r in range(world) if D.shard_range(n, r, world)[1] > D.shard_range(n, r, world)[0])
    assert sum(r[2] for r in res) == holders - 1
    merged = {}
    for part in rank0[3]:
        merged.update(part)
    assert sorted(merged) == sorted(["ep2_lb", "ep1_fb", "b4", "b3", "b2", "b1", "bi0"])
    assert all(merged[name][2] == rank0[4][name] for name in merged)
    if world >= 3:
        assert list(rank0[4].values()).count(0) <= 7 // world + 1


def test_wire_format_roundtrip_and_rejects_garbage():
    """distributed.encode_obj / decode_obj: what send_obj / recv_obj put on the wire (no pickle: decoding builds plain data only)"""
    import importlib
    D = importlib.import_module("zk-light-client-implementation_amd.distributed")
    triple = ({"config": {"num_wires": 135}, "gates": ["NoopGate"]}, {"circuit_digest": {"elements": [1, 2, 3, 4]}}, bytes(range(256)) * 500)
    obj = {"headers": {"b4": triple, "b3": ({"k": 1}, {}, {"public_inputs": [1, 2], "proof": {"wires_cap": [[1, 2, 3, 4]]}})},
           "ks": None, "t": (1, (2, b"x"), [3.5, True])}
    raw = D.encode_obj(obj)
    assert D.decode_obj(raw) == obj
    assert len(raw) < len(triple[2]) + 1000                      # byte strings travel raw, not hex / base64
    for bad in (raw[:-1], raw + b"\0", raw[:50]):
        with pytest.raises(Exception):
            D.decode_obj(bad)
    with pytest.raises(TypeError):
        D.encode_obj({"f": lambda: 0})


def _ok_worker(rank, world, port, failing, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    PL = importlib.import_module("zk-light-client-implementation_amd.pipeline")
    D = importlib.import_module("zk-light-client-implementation_amd.distributed")
    # the strong form's checkpoint on a stand-in pipeline: the rank in `failing` has an error, every rank must raise -- the failing one
    # its own exception, the others RemoteRankFailed -- after setting the exception on the futures its threads wait on
    p = object.__new__(PL.BlockPipeline)
    p.rank, p.world, p.comm_device, p.nthreads, p._sig_failed = rank, world, None, 2, False
    st = p._new_state([(b"msg", [], [])], True)
    if rank in failing:
        p._fail(st, ValueError("rank %d broke" % rank))
    try:
        p._strong_checkpoint(st, [])
        out = "passed"
    except D.RemoteRankFailed:
        out = "remote"
    except ValueError:
        out = "own"
    fut = st["sets"][0].future
    q.put((rank, out, fut.done() and fut.exception() is not None, D.all_ok(True), D.all_ok(rank != 0)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,failing", [(2, [1]), (3, [0]), (2, [])])
def test_a_failed_rank_fails_the_block_on_every_rank(world, failing):
    """in the strong form a rank's failure must reach the others BEFORE any exchange (pipeline._strong_checkpoint over
    distributed.all_ok, a MIN all-reduce): nobody folds a partial aggregate, nobody waits for a proof that will not come"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ok_worker, args=(r, world, port, failing, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, fut_failed, ok_all, ok_some in res:
        if not failing:
            assert out == "passed" and not fut_failed
        else:
            assert out == ("own" if rank in failing else "remote") and fut_failed
        assert ok_all is True and ok_some is False
