#!/bin/bash
# round 3, GPU call d: after the v_mad_i64_i32 fix (opaque limbs): fp ubench, MSM variants, counter list, SQ counters of both
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
tools/ubench/fp_ubench > gpurun_out/r03d_fp_ubench.txt 2>&1; cat gpurun_out/r03d_fp_ubench.txt
timeout 300 python tools/msm_quickbench.py 20 22 --variants=2p,1p,4p,2n > gpurun_out/r03d_msm_quick.txt 2>&1; grep -A1 "MSM" gpurun_out/r03d_msm_quick.txt
rocprofv3 -L > gpurun_out/r03d_counters.txt 2>&1; grep -c "SQ_" gpurun_out/r03d_counters.txt
rm -rf gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o msm -- python tools/msm_quickbench.py 22 > gpurun_out/r03d_msm_prof.log 2>&1
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03d_msm_2p22_kernel_stats.csv && head -8 gpurun_out/r03d_msm_2p22_kernel_stats.csv | cut -c1-140
f=$(find gpurun_out/prof_tmp -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && grep "bucket_sum" "$f" | head -2 > gpurun_out/r03d_bucket_trace_rows.txt; head -1 "$f" >> gpurun_out/r03d_bucket_trace_rows.txt
rm -rf gpurun_out/prof_tmp
tools/pmc_sq.sh r03d_msm python tools/msm_quickbench.py 22
head -4 gpurun_out/r03d_msm_pmc_sq.csv
tools/pmc_sq.sh r03d_fpub tools/ubench/fp_ubench
cat gpurun_out/r03d_fpub_pmc_sq.csv
