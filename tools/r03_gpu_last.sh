#!/bin/bash
# round 3, closing check on the final tree: the GPU test files not re-run since the NTT load batching (everything but the three
# multi-minute ones, which the full run of profiles/r03_pytest_gpu_final.log covered)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 270 python -m pytest tests/test_gpu_recursion.py tests/test_gpu_sha256.py tests/test_gpu_witness.py tests/test_gpu_bn254.py tests/test_gpu_ed25519.py -x -q > gpurun_out/r03_last_pytest.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r03_last_pytest.log
