#!/bin/bash
# round 3, GPU call j: two-pass Fr NTT
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bn254.py tests/test_gpu_groth16.py -x -q -m gpu > gpurun_out/r03j_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03j_pytest.log
timeout 300 python tools/bn254_quickbench.py 16 22 64 > gpurun_out/r03j_bn254_quick.txt 2>&1; grep "NTT\|MSM" gpurun_out/r03j_bn254_quick.txt
ZKLC_FR_NTT=stages timeout 300 python tools/bn254_quickbench.py 16 22 64 2>&1 | grep "NTT" | sed 's/^/stage path: /'
rm -rf gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o ntt -- python tools/bn254_quickbench.py 16 22 64 > gpurun_out/r03j_prof.log 2>&1
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03j_bn254_quick_kernel_stats.csv
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03j_bn254_quick_kernel_stats.csv')):
    if 'frn' in r['Name'] or 'msm_slice' in r['Name']: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e6, 3))
PY
rm -rf gpurun_out/prof_tmp
