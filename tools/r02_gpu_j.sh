#!/bin/bash
# round 2, call j: lazy Poseidon partial rounds + 2^13 NTT tile -- parity, kernel timings, per-kernel split of an Ed25519-circuit proof
set -u
TAG=${1:-r02j}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_goldilocks.py -m gpu -x -q > gpurun_out/${TAG}_pytest_gl.log 2>&1; echo "pytest goldilocks rc=$?"; tail -3 gpurun_out/${TAG}_pytest_gl.log
timeout 300 python tools/gl_quickbench.py > gpurun_out/${TAG}_quick.txt 2>&1; echo "quick rc=$?"; cat gpurun_out/${TAG}_quick.txt
ZKLC_NTT_TILE=12 timeout 300 python tools/gl_quickbench.py 2>&1 | grep -E "LDE|iNTT" > gpurun_out/${TAG}_quick_tile12.txt; echo "--- tile 12"; cat gpurun_out/${TAG}_quick_tile12.txt
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o prove -- python tools/prove_profile_ed25519.py 3 > gpurun_out/${TAG}_prove_profile.log 2>&1; echo "profile rc=$?"
tail -2 gpurun_out/${TAG}_prove_profile.log | cut -c1-600
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv; head -8 gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/prof_tmp
timeout 900 python -m pytest tests/test_gpu_plonky2.py tests/test_gpu_recursion.py -m gpu -x -q -k "not full_block" > gpurun_out/${TAG}_pytest_p2.log 2>&1; echo "pytest plonky2 rc=$?"; tail -3 gpurun_out/${TAG}_pytest_p2.log
