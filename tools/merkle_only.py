#!/usr/bin/env python3
"""the C3 Merkle commit alone (2^20 leaves x 234 columns, cap height 4), a few launches -- the target of the PMC passes that feed
bench.py's roofline block (tools/pmc_merkle.sh)"""
import sys
sys.path.insert(0, ".")
import torch
import zklc_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with zklc_amd.Context(0) as c:
    st = torch.cuda.Stream()
    N, batch = 1 << 20, 234
    g = torch.Generator(device="cuda").manual_seed(5)
    lde = torch.randint(0, 2**63 - 1, (batch, N), generator=g, device="cuda", dtype=torch.int64)
    tree = torch.empty(c.gl_merkle_tree_words(20, 4), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(reps):
        c.gl_merkle_commit_dev(lde, N, 20, batch, 4, tree, stream=st)
    torch.cuda.synchronize()
print("done")
