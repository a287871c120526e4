#!/bin/bash
# round 3, GPU call h: MSM with slice-based bucket accumulation
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/msm_quickbench.py 16 20 22 --variants=2,1,3 > gpurun_out/r03h_msm_quick.txt 2>&1; grep -A1 "MSM" gpurun_out/r03h_msm_quick.txt
rm -rf gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o msm -- python tools/msm_quickbench.py 22 > gpurun_out/r03h_msm_prof.log 2>&1
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03h_msm_2p22_kernel_stats.csv
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03h_msm_2p22_kernel_stats.csv')):
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e6, 3))
PY
rm -rf gpurun_out/prof_tmp
timeout 600 python -m pytest tests/test_gpu_bn254.py tests/test_gpu_groth16.py -x -q -m gpu > gpurun_out/r03h_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03h_pytest.log
