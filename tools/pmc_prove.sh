#!/bin/bash
# HBM traffic of the plonky2 prover kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE do not fit one pass)
# over tools/prove_profile.py (synthetic 2^17 x 234 circuit, 2 proofs).  Output: gpurun_out/<tag>_pmc_{fetch,write}.csv
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_tmp -o pmc -- python tools/prove_profile.py ed 17 2 > gpurun_out/${TAG}_pmc_$c.log 2>&1
  f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1)
  python - "$f" "$c" > gpurun_out/${TAG}_pmc_${c}.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if r.get("Counter_Name") != sys.argv[2]:
        continue
    k = r["Kernel_Name"].split("(")[0]
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
print("kernel,launches,%s_total_KiB,%s_per_launch_KiB" % (sys.argv[2], sys.argv[2]))
for k, (n, v) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%s,%d,%.0f,%.1f" % (k, n, v, v / n))
PY
done
rm -rf gpurun_out/pmc_tmp
head -12 gpurun_out/${TAG}_pmc_FETCH_SIZE.csv; head -8 gpurun_out/${TAG}_pmc_WRITE_SIZE.csv
