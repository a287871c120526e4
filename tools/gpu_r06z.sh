#!/bin/bash
# round 6, GPU job z: per-kernel summary of the three-stream Groth16 prove at 2^22 (plain form)
set -u
TAG=${1:-r06z}; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_tmp
MODES=0 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o g16 -- python tools/groth16_quickbench.py 22 3 > gpurun_out/${TAG}_groth16_rocprof.log 2>&1; echo "rc=$?"
find gpurun_out/prof_tmp -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_groth16_2p22_kernel_stats.csv \;
rm -rf gpurun_out/prof_tmp
grep "groth16 prove" gpurun_out/${TAG}_groth16_rocprof.log
head -12 gpurun_out/${TAG}_groth16_2p22_kernel_stats.csv | cut -c1-160
