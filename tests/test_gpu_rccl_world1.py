"""RCCL on the ONE GPU of the test box: `torch.distributed` with backend "nccl" (= RCCL on ROCm) and world_size 1.
The collectives of the multi-GPU path -- the all-gather of `msm_sharded`, the two all-gathers + MAX all-reduce of `gather_bytes`,
the MIN all-reduce of `all_ok`, the barrier / MAX all-reduce that brackets bench.py's timed region -- execute for real on device
tensors through librccl; a world of one has no peer, so the point-to-point transfers of `tree_fold` / `gather_objects`
(send_obj / recv_obj) cannot run here: tests/test_gpu_multi.py covers them on a box with two GPUs, the gloo tests on the CPU.
Runs in a child process so that the process group never leaks into the test session."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import importlib, json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
import zklc_amd
from oracle import cport
D = importlib.import_module("zk-light-client-implementation_amd.distributed")
out = {}
n = 1 << 12
pts = cport.bn254_gen_points(n, 3, 5)
rng = np.random.default_rng(7)
sc = rng.integers(0, 2**62, size=(n, 4), dtype=np.uint64)
sc[:, 3] >>= np.uint64(4)
with zklc_amd.Context(0) as ctx:
    calls = []
    def f(p, s):
        calls.append(len(p))
        return ctx.bn254_g1_msm(np.ascontiguousarray(p), np.ascontiguousarray(s))
    got, inf = D.msm_sharded(f, f, pts, sc, device=dev)
want, winf, _ = cport.bn254_msm(pts, sc, nthreads=4)
out["msm_equal"] = bool(inf == winf and np.asarray(got).tolist() == want.tolist())
out["msm_calls"] = calls                      # [n, 1]: the local MSM, then the unit-scalar MSM over the ALL-GATHERED partial
chunks = [bytes([i]) * (1000 * (i + 1)) for i in range(3)]
out["gather_bytes_equal"] = D.gather_bytes(chunks, device=dev) == [chunks]
out["all_ok"] = [D.all_ok(True, device=dev), D.all_ok(False, device=dev)]
# the bracket of bench.py's timed region: barrier + synchronize, MAX of the elapsed time over the ranks (on the device with RCCL)
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([1.25, 3.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
c = torch.tensor([41], device=dev, dtype=torch.int64)
dist.all_reduce(c, op=dist.ReduceOp.SUM)
out["reduce"] = [float(t[0]), float(t[1]), int(c[0])]
out["comm_device"] = str(D._comm_device(dev))
dist.destroy_process_group()
print("RCCL1 " + json.dumps(out))
'''


def test_collectives_of_the_multi_gpu_path_over_rccl_world1():
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = json.loads(next(ln for ln in r.stdout.splitlines() if ln.startswith("RCCL1 "))[6:])
    assert j["msm_equal"] and j["msm_calls"] == [1 << 12, 1]
    assert j["gather_bytes_equal"] and j["all_ok"] == [True, False]
    assert j["reduce"] == [1.25, 3.5, 41] and j["comm_device"] == "cuda:0"
