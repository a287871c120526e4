#!/bin/bash
# round 6, GPU job u: where are the prover threads while their hardware queues are idle?  20 ms thread samples of 8 timed blocks
set -u
TAG=r06u; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_warm_detail.json timeout 900 $B > gpurun_out/${TAG}_warm_line.json 2> gpurun_out/${TAG}_warm.err; echo "warm rc=$?"
ZKLC_BENCH_SAMPLE=gpurun_out/${TAG}_samples.jsonl ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_sampled_detail.json timeout 900 $B > gpurun_out/${TAG}_sampled_line.json 2> gpurun_out/${TAG}_sampled.err; echo "sampled rc=$?"
python tools/thread_samples.py gpurun_out/${TAG}_samples.jsonl > gpurun_out/${TAG}_thread_samples.txt 2>&1
head -60 gpurun_out/${TAG}_thread_samples.txt | cut -c1-420
gzip -f gpurun_out/${TAG}_samples.jsonl
