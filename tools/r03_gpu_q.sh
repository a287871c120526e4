#!/bin/bash
# round 3, GPU call q: FETCH_SIZE + SQ counters of the quotient kernels of the Ed25519-circuit proof (is U32AddMany bandwidth-bound?)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_tmp -o pmc -- python tools/prove_profile_ed25519.py 1 > gpurun_out/r03q.log 2>&1; echo "rc=$?"
f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1); cp "$f" gpurun_out/r03q_pmc_prove.csv
rm -rf gpurun_out/pmc_tmp
python - <<'PY'
import csv, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/r03q_pmc_prove.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "quotient" in k or "ntt_pass" in k or "hash_leaves" in k or "fri_combine" in k:
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in per.items():
    print(k[:44].ljust(44), {c[:14]: round(max(x) / 1e6, 1) for c, x in d.items()}, "n", len(next(iter(d.values()))))
PY
