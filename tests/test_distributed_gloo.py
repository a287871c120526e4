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
