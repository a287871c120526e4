#!/bin/bash
# round 3, GPU call l: BASELINE configs[4] (C5) at its stated size -- a synthetic epoch of 1000 validators through the block pipeline
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python bench.py --c5-validators 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-bn254-extras > gpurun_out/r03l_bench_c5_1000.json 2> gpurun_out/r03l_bench_c5_1000.err
echo "rc=$?"; tail -c 1500 gpurun_out/r03l_bench_c5_1000.json; tail -5 gpurun_out/r03l_bench_c5_1000.err
