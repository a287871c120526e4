"""Quick timing of the Goldilocks kernels at the C3 shapes (not the driver's bench)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zklc_amd  # noqa: E402

P = 2**64 - 2**32 + 1
OUT_BR = 4


def rand_gl(shape, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randint(0, 2**63 - 1, shape, generator=g, device="cuda", dtype=torch.int64)
    return a  # < 2^63 < p: canonical


def timeit(fn, st, iters=5):
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
    for log_n, rate, batch in [(18, 3, 234), (17, 3, 234), (17, 3, 20), (12, 3, 135)]:
        n, N = 1 << log_n, 1 << (log_n + rate)
        coeffs = rand_gl((batch, n), 1)
        out = torch.empty((batch, N), dtype=torch.int64, device="cuda")
        ms = timeit(lambda: c.gl_lde_dev(coeffs, log_n, rate, batch, 7, out, flags=OUT_BR, stream=st), st)
        alg = 8 * (n + N) * batch
        bf = (N // 2) * (log_n + rate) * batch
        print("LDE 2^%d->2^%d x%d: %.3f ms  %.1f GB/s (algorithmic)  %.2f Gbutterfly/s" % (log_n, log_n + rate, batch, ms, alg / ms / 1e6, bf / ms / 1e6), flush=True)
        vals = rand_gl((batch, n), 2)
        ms = timeit(lambda: c.gl_ntt_dev(vals, log_n, batch, flags=1 | OUT_BR, stream=st), st)
        print("iNTT 2^%d x%d (bitrev out): %.3f ms  %.1f GB/s" % (log_n, batch, ms, 16 * n * batch / ms / 1e6), flush=True)
        if batch in (234, 135):
            words = c.gl_merkle_tree_words(log_n + rate, 4)
            tree = torch.empty(words, dtype=torch.int64, device="cuda")
            ms = timeit(lambda: c.gl_merkle_commit_dev(out, N, log_n + rate, batch, 4, tree, stream=st), st, iters=3)
            perms = N * ((batch + 7) // 8) + N
            print("Merkle commit 2^%d leaves x width %d cap 4: %.3f ms  %.1f Mleaf/s  %.1f Mperm/s  %.1f GB/s (8w+32 B/leaf)" % (
                log_n + rate, batch, ms, N / ms / 1e3, perms / ms / 1e3, (8 * batch + 32) * N / ms / 1e6), flush=True)
    ns = 1 << 22
    states = rand_gl((ns, 12), 3)
    ms = timeit(lambda: c.poseidon_gl_permute_dev(states, ns, stream=st), st)
    print("Poseidon permute x2^22: %.3f ms  %.1f Mperm/s" % (ms, ns / ms / 1e3), flush=True)
