"""Quick timing of the Ed25519 verify kernel at several batch sizes (not the driver's bench).  The four variants of round 1
(profiles/r01_ed25519_variants_v2.txt) are gone: only the fastest (v1) is compiled."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zklc_amd  # noqa: E402
from tests.edcases import synthetic_set  # noqa: E402

base_n = 256
pks, sigs, msg = synthetic_set(base_n, seed=4)
for n in [100, 1 << 14, 1 << 18, 1 << 20]:
    reps = (n + base_n - 1) // base_n
    pk = torch.tensor(np.frombuffer(b"".join(pks), np.uint8).copy(), device="cuda").view(base_n, 32).repeat(reps, 1)[:n].contiguous()
    sg = torch.tensor(np.frombuffer(b"".join(sigs), np.uint8).copy(), device="cuda").view(base_n, 64).repeat(reps, 1)[:n].contiguous()
    m = torch.tensor(np.frombuffer(msg, np.uint8).copy(), device="cuda")
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    for v in (1,):
        with zklc_amd.Context(0) as c:
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            c.ed25519_verify_batch_dev(pk, sg, m, len(msg), 0, n, ok, stream=st)
            torch.cuda.synchronize()
            assert int(ok.sum()) == n
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 3 if n >= (1 << 18) else 10
            e0.record(st)
            for _ in range(iters):
                c.ed25519_verify_batch_dev(pk, sg, m, len(msg), 0, n, ok, stream=st)
            e1.record(st)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            print("n=%8d variant=%d  %.3f ms  %.3f Msig/s" % (n, v, ms, n / ms / 1e3), flush=True)
