#!/usr/bin/env python3
"""the coset LDE alone (234 polynomials, 2^LOGN -> 2^(LOGN+3), bit-reversed output), a few launches -- the target of the PMC passes
that feed bench.py's stages.lde roofline block (tools/pmc_lde.sh).  LOGN = argv[2], default 17 (C3); 18 = the product shape of the
Ed25519 circuit's commitments (prove_crypto/ed25519.rs:60)."""
import sys
sys.path.insert(0, ".")
import torch
import zklc_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with zklc_amd.Context(0) as c:
    st = torch.cuda.Stream()
    log_n, rate, batch = (int(sys.argv[2]) if len(sys.argv) > 2 else 17), 3, 234
    g = torch.Generator(device="cuda").manual_seed(0xC0FFEE)
    coeffs = torch.randint(0, 2**63 - 1, (batch, 1 << log_n), generator=g, device="cuda", dtype=torch.int64)
    out = torch.empty((batch, 1 << (log_n + rate)), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(reps):
        c.gl_lde_dev(coeffs, log_n, rate, batch, 7, out, flags=zklc_amd._lib.NTT_OUT_BITREV, stream=st)
    torch.cuda.synchronize()
print("done")
