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
