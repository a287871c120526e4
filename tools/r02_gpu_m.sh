#!/bin/bash
# round 2, call m: batched wire loads + scalar table loads in the quotient kernels
set -u
TAG=${1:-r02m}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_plonky2.py -m gpu -x -q -k "bit_for_bit or ed25519_circuit" > gpurun_out/${TAG}_pytest_p2.log 2>&1; echo "pytest plonky2 rc=$?"; tail -3 gpurun_out/${TAG}_pytest_p2.log
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o prove -- python tools/prove_profile_ed25519.py 3 > gpurun_out/${TAG}_prove_profile.log 2>&1; echo "profile rc=$?"
grep -n "wires_commit" gpurun_out/${TAG}_prove_profile.log | cut -c1-400
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv; head -14 gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv | cut -c1-60,100-230
rm -rf gpurun_out/prof_tmp
ZKLC_P2_GATE_LAUNCH=single timeout 600 python tools/prove_profile_ed25519.py 3 2>&1 | grep wires_commit | cut -c1-400 > gpurun_out/${TAG}_prove_single.txt; echo "--- one launch per gate:"; cat gpurun_out/${TAG}_prove_single.txt
