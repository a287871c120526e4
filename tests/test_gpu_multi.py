"""N > 1 on real devices: one process per GPU over RCCL (torch.distributed backend "nccl").  These tests need at least
two GPUs and SKIP on a one-GPU box (the gloo tests of tests/test_distributed_gloo.py cover the same logic on the CPU with the
oracle standing in for the kernels); on a multi-GPU node they are the first thing to run:
    python -m pytest tests/test_gpu_multi.py -m gpu -q
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


needs2 = pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (RCCL over xGMI); the gloo tests cover the logic on the CPU")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _msm_worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import importlib
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import zklc_amd
    from oracle import cport
    D = importlib.import_module("zk-light-client-implementation_amd.distributed")
    pts = cport.bn254_gen_points(n, 3, 5)
    rng = np.random.default_rng(7)
    sc = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] >>= np.uint64(4)
    lo, hi = D.shard_range(n, rank, world)
    dev = torch.device("cuda", rank)
    with zklc_amd.Context(rank) as ctx:
        f = lambda p, s: ctx.bn254_g1_msm(np.ascontiguousarray(p), np.ascontiguousarray(s))
        out, inf = D.msm_sharded(f, f, pts[lo:hi], sc[lo:hi], device=dev)
        # the binary-tree fold over the ranks with byte payloads (what a proof triple is on the wire)
        local = ({"rank": rank}, {"vd": [rank]}, bytes([rank]) * 1000)
        total = D.tree_fold(local, lambda a, b: ({"rank": a[0]["rank"]}, a[1], a[2] + b[2]), device=dev)
    q.put((rank, np.asarray(out).tolist(), bool(inf), None if total is None else len(total[2])))
    dist.destroy_process_group()


@needs2
def test_msm_sharded_over_rccl_equals_one_gpu():
    """distributed.msm_sharded on two devices (index shards, all-gather of the partial sums, local addition) == the oracle's MSM of
    the whole instance; the tree fold's point-to-point transfers (distributed.send_obj / recv_obj) over RCCL"""
    import torch.multiprocessing as mp
    from oracle import cport
    world, n = 2, 1 << 12
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_msm_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
    pts = cport.bn254_gen_points(n, 3, 5)
    rng = np.random.default_rng(7)
    sc = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] >>= np.uint64(4)
    want, winf, _ = cport.bn254_msm(pts, sc, nthreads=4)
    for rank, out, inf, folded in res:
        assert inf == winf and out == want.tolist()
    assert res[0][3] == 2000 and res[1][3] is None


@needs2
def test_strong_scaling_block_over_rccl():
    """bench.py with two ranks over RCCL: the weak headline, then ONE block over both ranks (signature shards, tree fold, header proofs
    on rank 1) -- bench.py verifies the final proofs itself and reports `block_i.strong.final_proof_verified`.  ~10 minutes (every
    rank builds the circuits): ZKLC_SLOW_TESTS only."""
    import json
    if not os.environ.get("ZKLC_SLOW_TESTS"):
        pytest.skip("slow: two ranks build every circuit of the block DAG (set ZKLC_SLOW_TESTS=1)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-bn254-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=3000, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = next(ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"'))
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["final_proof_verified"] and j["block_i"]["strong"]["final_proof_verified"]
    assert j["stages"]["msm"]["strong"]["equal"] is True          # compact line: sharded MSM == the single-GPU point


def test_two_ranks_share_the_one_gpu_over_gloo():
    """The multi-rank bench path kept alive on a ONE-GPU box (test hygiene): `python bench.py --gpus 2 --backend gloo`
    starts its own two ranks (no torchrun), both on cuda:0 -- RCCL cannot put two ranks on one device, gloo can: collectives on host
    tensors, kernels on the shared GPU -- and runs the weak step (each rank its own block) AND the strong section (ONE block over
    both ranks: signature shards, local folds, the tree fold's point-to-point exchange, header proofs on rank 1, joins and the wrap
    on rank 0) with the final proofs verified by the oracle's verifier inside bench.py.  Several minutes (two ranks build or load
    every circuit); ZKLC_FAST_TESTS=1 skips."""
    import json
    if os.environ.get("ZKLC_FAST_TESTS"):
        pytest.skip("ZKLC_FAST_TESTS: a multi-minute two-rank bench run")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ZKLC_BENCH_DETAIL=os.path.join(ROOT, "gpurun_out", "test_two_ranks_detail.json"))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    import torch
    torch.cuda.empty_cache()                   # this process's cached blocks are HBM the two ranks cannot have (r06m: the test failed
    # in a full-suite run with three witness buffers per rank and passed alone; ranks that share a device now size for the share)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-bn254-extras", "--c5-validators", "0"], capture_output=True, text=True, timeout=1500, env=env)
    if r.returncode != 0:                      # the ranks' tracebacks are long: the whole stderr goes to a file, the first error here
        with open(os.path.join(ROOT, "gpurun_out", "test_two_ranks_stderr.log"), "w") as f:
            f.write(r.stderr)
        first = r.stderr.find("Traceback")
        raise AssertionError(r.stderr[max(0, first):first + 3000] if first >= 0 else r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["final_proof_verified"] is True
    blk = line["block_i"]
    assert blk["blocks_checked"] == 1 and blk["strong"]["final_proof_verified"] is True
    assert line["stages"]["msm"]["strong"]["equal"] is True          # the index-sharded MSM == the single-GPU point
    print("two ranks on one GPU: weak %.2f s per step, strong %.2f s per block" % (line["ms_per_step"] / 1e3, blk["strong"]["seconds_per_block"]))
