#!/bin/bash
# SQ counter pass (VALU issue / busy / stall) of a command's kernels: rocprofv3 --pmc only (no other trace domains).
#   tools/pmc_sq.sh <tag> <command...>     -> gpurun_out/<tag>_pmc_sq.csv  (per kernel: launches, counters per launch)
set -u
TAG=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES \
   --output-format csv -d gpurun_out/pmc_tmp -o pmc -- "$@" > gpurun_out/${TAG}_pmc_sq.log 2>&1
echo "rocprof pmc rc=$?"
f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1)
python - "$f" > gpurun_out/${TAG}_pmc_sq.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
names = []
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    agg[k][c] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
print("kernel,launches," + ",".join(n + "_per_launch" for n in names))
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    n = len(cnt[k])
    print(k + "," + str(n) + "," + ",".join("%.0f" % (agg[k][c] / n) for c in names))
PY
rm -rf gpurun_out/pmc_tmp
head -14 gpurun_out/${TAG}_pmc_sq.csv
