#!/bin/bash
# round 3, GPU call b: MSM v2 -- parity tests, A/B of the bucket-kernel variants, per-kernel split, SQ counters
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_bn254.py tests/test_gpu_groth16.py -x -q -m gpu > gpurun_out/r03b_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03b_pytest.log
timeout 300 python tools/msm_quickbench.py 16 20 22 --variants=2p,1p,3p,4p,2n,4n > gpurun_out/r03b_msm_quick.txt 2>&1; echo "quick rc=$?"; grep MSM gpurun_out/r03b_msm_quick.txt
rm -rf gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o msm -- python tools/msm_quickbench.py 22 > gpurun_out/r03b_msm_prof.log 2>&1
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03b_msm_2p22_kernel_stats.csv && head -14 gpurun_out/r03b_msm_2p22_kernel_stats.csv
rm -rf gpurun_out/prof_tmp
tools/pmc_sq.sh r03b_msm python tools/msm_quickbench.py 22
head -8 gpurun_out/r03b_msm_pmc_sq.csv
for pass in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_tmp
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -o pmc -- python tools/msm_quickbench.py 22 > gpurun_out/r03b_msm_pmc_$pass.log 2>&1
  f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1)
  python - "$f" $pass > gpurun_out/r03b_msm_pmc_$pass.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg, cnt = collections.defaultdict(float), collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    agg[k] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
print("kernel,launches,%s_per_launch" % sys.argv[2])
for k in sorted(agg, key=lambda k: -agg[k]):
    print("%s,%d,%.0f" % (k, len(cnt[k]), agg[k] / len(cnt[k])))
PY
  head -6 gpurun_out/r03b_msm_pmc_$pass.csv
done
rm -rf gpurun_out/pmc_tmp
