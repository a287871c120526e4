#!/usr/bin/env python3
"""Reproducer for the round-5 bench hang (profiles/r05h_bench_hang_with_blocking_wait_default.txt): a process that has ALREADY run GPU
work through PyTorch flips the device's wait mode with hipSetDeviceFlags(hipDeviceScheduleBlockingSync) -- what zklc_init did under
ZKLC_BLOCKING_WAIT=1 -- keeps working, and then calls torch.cuda.empty_cache().  Prints one line per step with its wall time; run it
under `timeout 120` (a hang is the finding).  With `--no-flip` the same steps without the flag.   python tools/repro_blocking_sync_hang.py"""
import ctypes
import sys
import threading
import time

import torch

t0 = time.perf_counter()
say = lambda s: print("%7.2f s  %s" % (time.perf_counter() - t0, s), flush=True)
x = torch.randn(1 << 24, device="cuda")
(x * 2).sum().item()
say("torch kernels ran on the device (primary context active, default wait mode)")
hip = ctypes.CDLL([m.split()[-1] for m in open("/proc/self/maps") if "libamdhip64" in m][0])
if "--no-flip" not in sys.argv:
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(0x4))            # hipDeviceScheduleBlockingSync
    say("hipSetDeviceFlags(hipDeviceScheduleBlockingSync) -> %d" % rc)
streams = [torch.cuda.Stream() for _ in range(4)]


def work(s):
    with torch.cuda.stream(s):
        for _ in range(200):
            y = torch.empty(1 << 22, device="cuda")
            y.normal_()
            (y * y).sum()
        s.synchronize()


ths = [threading.Thread(target=work, args=(s,)) for s in streams]
[t.start() for t in ths]
[t.join() for t in ths]
say("4 threads x 200 allocations / kernels on their own streams, synchronised")
del x
torch.cuda.synchronize()
say("torch.cuda.synchronize()")
torch.cuda.empty_cache()
say("torch.cuda.empty_cache() returned")
