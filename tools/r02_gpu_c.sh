#!/bin/bash
set -u
TAG=${1:-r02c}
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/ubench/valu_ubench > gpurun_out/${TAG}_valu_ubench.txt 2>&1; tail -12 gpurun_out/${TAG}_valu_ubench.txt
tools/pmc_sq.sh ${TAG} python tools/gl_quickbench.py
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o prove -- python tools/prove_profile_ed25519.py 3 > gpurun_out/${TAG}_prove_ed25519.log 2>&1; echo "rocprof rc=$?"
find gpurun_out/prof_tmp -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv \;
rm -rf gpurun_out/prof_tmp
tail -3 gpurun_out/${TAG}_prove_ed25519.log
head -12 gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv | cut -c1-150
