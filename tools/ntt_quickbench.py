"""Timing of the Goldilocks LDE / NTT passes alone at the prover's shapes (A/B of the pass kernels: ZKLC_NTT_R8=1, ZKLC_NTT_RADIX2=1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zklc_amd  # noqa: E402

OUT_BR = 4


def rand_gl(shape, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randint(0, 2**63 - 1, shape, generator=g, device="cuda", dtype=torch.int64)


def timeit(fn, st, iters=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


with zklc_amd.Context(0) as c:
    st = torch.cuda.Stream()
    for log_n, rate, batch in [(18, 3, 234), (17, 3, 234), (14, 3, 135), (12, 3, 135)]:
        n, N = 1 << log_n, 1 << (log_n + rate)
        coeffs = rand_gl((batch, n), 1)
        out = torch.empty((batch, N), dtype=torch.int64, device="cuda")
        ms = timeit(lambda: c.gl_lde_dev(coeffs, log_n, rate, batch, 7, out, flags=OUT_BR, stream=st), st)
        bf = (N // 2) * (log_n + rate) * batch
        print("LDE 2^%d->2^%d x%d: %.3f ms  %.1f GB/s (algorithmic)  %.2f Gbutterfly/s" % (
            log_n, log_n + rate, batch, ms, 8 * (n + N) * batch / ms / 1e6, bf / ms / 1e6), flush=True)
        vals = rand_gl((batch, n), 2)
        ms = timeit(lambda: c.gl_ntt_dev(vals, log_n, batch, flags=1 | OUT_BR, stream=st), st)
        print("iNTT 2^%d x%d (bitrev out): %.3f ms" % (log_n, batch, ms), flush=True)
