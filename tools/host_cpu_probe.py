#!/usr/bin/env python3
"""host CPU seconds against wall seconds of a proving thread (is the host spinning while the GPU works?): a few proofs of the
synthetic Ed25519 shape on one stream, then the same with the thread's per-call breakdown from /proc/self/task
   python tools/host_cpu_probe.py [bits] [reps]"""
import os
import sys
import time
sys.path.insert(0, ".")
import zklc_amd
from zklc_amd.plonky2 import synthetic as SY, wide_ecc_config, HASH_GL
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 17
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = zklc_amd.Context(0)
cfg = wide_ecc_config()
data, wires, pis = SY.synthetic_circuit(bits, cfg, SY.ed25519_shape_mix(cfg), num_public_inputs=584, seed=1)
prover = data.prover(ctx, HASH_GL)
prover.prove_bytes(wires, pis)
import torch
d_w = torch.from_numpy(wires.view("int64")).to("cuda:0")
prover.prove_dev(d_w.data_ptr(), pis, stream=ctx.stream_ptr())


def tasks():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read().rsplit(")", 1)[1].split()
            out[t] = (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK")
        except OSError:
            pass
    return out


t0, c0, k0 = time.perf_counter(), time.process_time(), tasks()
for _ in range(reps):
    prover.prove_dev(d_w.data_ptr(), pis, stream=ctx.stream_ptr())
t1, c1, k1 = time.perf_counter(), time.process_time(), tasks()
print("2^%d x 234, %d proofs: wall %.3f s, process CPU %.3f s (%.2f cores busy)" % (bits, reps, t1 - t0, c1 - c0, (c1 - c0) / (t1 - t0)))
print("per thread CPU s:", {t: round(k1[t] - k0.get(t, 0), 3) for t in k1 if k1[t] - k0.get(t, 0) > 0.005})
print("stage ms:", {k: round(v, 2) for k, v in prover.last_timings().items()})
