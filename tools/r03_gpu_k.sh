#!/bin/bash
# round 3, GPU call k: MSM with the quad-shared final doublings (parity + timing + kernel stats), then the PMC passes of the MSM,
# the coset LDE and the Merkle leaf kernel (-> profiles/*_pmc_latest.json)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bn254.py -x -q > gpurun_out/r03k_pytest_bn254.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03k_pytest_bn254.log
timeout 300 python tools/msm_quickbench.py 16 20 22 > gpurun_out/r03k_msm_quick.txt 2>&1; cat gpurun_out/r03k_msm_quick.txt
rm -rf gpurun_out/prof_k
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k -o msm -- python tools/msm_quickbench.py 22 > /dev/null 2>&1
f=$(find gpurun_out/prof_k -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r03k_msm_2p22_kernel_stats.csv; head -20 "$f" | cut -c1-160
rm -rf gpurun_out/prof_k
bash tools/pmc_msm.sh r03k
bash tools/pmc_lde.sh r03k
bash tools/pmc_merkle.sh r03k
