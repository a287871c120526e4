#!/usr/bin/env python3
"""where the seconds of a cold start go: BlockPipeline() + the first prove_block_bft of the mainnet window under cProfile
   python tools/cold_start_profile.py [cache_dir]      (an empty / missing directory = a cold circuit cache)"""
import cProfile
import json
import os
import pstats
import resource
import sys
import time
sys.path.insert(0, ".")
os.environ["ZKLC_CIRCUIT_CACHE"] = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "/tmp/zklc_cold_cache"
import torch  # noqa: E402,F401
import zklc_amd  # noqa: E402,F401
from zklc_amd.pipeline import BlockPipeline, BlockWindow  # noqa: E402
win = BlockWindow.from_fixture(json.load(open(os.path.join("tests", "golden", "block_window_HPi5.json"))))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
pipe = BlockPipeline(0)
rep = pipe.prewarm(win) if "--no-prewarm" not in sys.argv else None
t1 = time.perf_counter()
res = pipe.prove_block_bft(win)
t2 = time.perf_counter()
pr.disable()
res2 = pipe.prove_block_bft(win)
t3 = time.perf_counter()
print("prewarm:", rep)
print("pipeline construction + prewarm %.1f s, first block %.1f s, second block %.1f s, peak RSS %.1f GB" % (
    t1 - t0, t2 - t1, t3 - t2, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))
print("dag counts:", res.dag_counts)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(60)
pipe.close()
