#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, bench line, rocprofv3 kernel stats.
# Everything lands under gpurun_out/ and is copied into profiles/ by hand afterwards.
set -u
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/${TAG}_bench.json
# kernel statistics: the headline + MSM/LDE/Merkle stages, and (separately) the kernels of plonky2 proofs
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-prove > gpurun_out/${TAG}_prof_bench.log 2>&1; echo "rocprof bench rc=$?"
find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_bench_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o prove -- python tools/prove_profile.py ed 17 5 > gpurun_out/${TAG}_prof_prove.log 2>&1; echo "rocprof prove rc=$?"
find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_prove_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_prof
tail -3 gpurun_out/${TAG}_pytest_gpu.log
