#!/usr/bin/env python3
"""device witness generation of the reference's Ed25519 circuit: timing per batch size; run under rocprofv3 --kernel-trace --stats
for the per-kernel split.   python tools/witness_profile.py [batch sizes ...]"""
import json
import os
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
import zklc_amd
from zklc_amd.plonky2 import CircuitBuilder, wide_ecc_config, ed25519_circuit as E
sizes = [int(x) for x in sys.argv[1:]] or [3, 16]
j = json.load(open(os.path.join("tests", "golden", "ed25519_near_c1_small.json")))
msg = bytes.fromhex(j["msg"])
b = CircuitBuilder(wide_ecc_config())
targets = E.ed25519_circuit(b, 8 * len(msg))
data = b.build()
fills = [E.fill_ecdsa_targets(targets, msg, bytes.fromhex(x["approval"])[2:], bytes.fromhex(x["validator_tail"])[1:33]) for x in j["entries"]]
data.witness_program(fills[0])
ctx = zklc_amd.Context(0)
dw = data.device_witness(ctx)
for k in sizes:
    batch = [fills[i % 3] for i in range(k)]
    vals = dw.input_matrix(batch)
    d = torch.zeros((k, dw.num_wires, dw.n_rows), dtype=torch.int64, device="cuda")
    dw.run(d.data_ptr(), input_values=vals)
    torch.cuda.synchronize()
    t0 = time.time()
    dw.run(d.data_ptr(), input_values=vals)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("batch %d: %s  %.3f s = %.1f ms per witness" % (k, dw.info(k), dt, dt / k * 1e3), flush=True)
    del d
