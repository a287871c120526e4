#!/bin/bash
# round 3, GPU call i: Poseidon-BN254 (lazy sums of products, four-lane permutation for the small trees): parity + wrap-shape timing
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bn254.py tests/test_gpu_recursion.py -x -q -m gpu > gpurun_out/r03i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03i_pytest.log
timeout 300 python tools/prove_quickbench.py 13 12 5 > gpurun_out/r03i_prove_quick.txt 2>&1; grep -A1 "recursion" gpurun_out/r03i_prove_quick.txt
ZKLC_BN254_COOP=0 timeout 300 python tools/prove_quickbench.py 13 12 5 > gpurun_out/r03i_prove_quick_nocoop.txt 2>&1; grep -A1 "BN128" gpurun_out/r03i_prove_quick_nocoop.txt
