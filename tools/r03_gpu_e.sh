#!/bin/bash
# round 3, GPU call e: 256-thread bucket workgroups; ubench with the fixed field code
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/ubench/fp_ubench > gpurun_out/r03e_fp_ubench.txt 2>&1; grep "ec_add\|mad" gpurun_out/r03e_fp_ubench.txt
timeout 300 python tools/msm_quickbench.py 20 22 --variants=2p,1p,3p,2n > gpurun_out/r03e_msm_quick.txt 2>&1; grep -A1 "MSM" gpurun_out/r03e_msm_quick.txt
tools/pmc_sq.sh r03e_msm python tools/msm_quickbench.py 22
head -3 gpurun_out/r03e_msm_pmc_sq.csv
