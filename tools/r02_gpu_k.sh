#!/bin/bash
# round 2, call k: carry-free constraint accumulation (p2_consumer) + table-driven FRI combine -- parity, per-kernel split, Poseidon counters
set -u
TAG=${1:-r02k}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_plonky2.py tests/test_gpu_recursion.py -m gpu -x -q -k "not full_block" > gpurun_out/${TAG}_pytest_p2.log 2>&1; echo "pytest plonky2 rc=$?"; tail -3 gpurun_out/${TAG}_pytest_p2.log
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o prove -- python tools/prove_profile_ed25519.py 3 > gpurun_out/${TAG}_prove_profile.log 2>&1; echo "profile rc=$?"
grep -n "wires_commit" gpurun_out/${TAG}_prove_profile.log | cut -c1-400
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv; head -22 gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv | cut -c1-60,100-230
rm -rf gpurun_out/prof_tmp
bash tools/pmc_merkle.sh ${TAG} 2>&1 | tail -2
