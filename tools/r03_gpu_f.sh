#!/bin/bash
# round 3, GPU call f: EC special cases out of line; per-kernel split; the other BN254 kernels after the v_mad_i64_i32 fix
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/msm_quickbench.py 16 20 22 --variants=2p,1p,3p > gpurun_out/r03f_msm_quick.txt 2>&1; grep -A1 "MSM" gpurun_out/r03f_msm_quick.txt
rm -rf gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o msm -- python tools/msm_quickbench.py 22 > gpurun_out/r03f_msm_prof.log 2>&1
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03f_msm_2p22_kernel_stats.csv && head -8 gpurun_out/r03f_msm_2p22_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof_tmp
timeout 300 python tools/bn254_quickbench.py > gpurun_out/r03f_bn254_quick.txt 2>&1; cat gpurun_out/r03f_bn254_quick.txt | tail -6
timeout 600 python -m pytest tests/test_gpu_bn254.py tests/test_gpu_groth16.py tests/test_gpu_ed25519.py -x -q -m gpu > gpurun_out/r03f_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03f_pytest.log
