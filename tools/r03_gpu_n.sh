#!/bin/bash
# round 3, GPU call n: shift-twiddle NTT passes vs the radix-8 kernel: timing and SQ / LDS counters of both
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/ntt_quickbench.py > gpurun_out/r03n_ntt_g4.txt 2>&1; cat gpurun_out/r03n_ntt_g4.txt | grep -v amdgpu.ids
ZKLC_NTT_R8=1 timeout 300 python tools/ntt_quickbench.py 2>&1 | grep -v amdgpu.ids | sed 's/^/r8: /' | tee gpurun_out/r03n_ntt_r8.txt
for v in g4 r8; do
  [ $v = r8 ] && export ZKLC_NTT_R8=1
  rm -rf gpurun_out/pmc_tmp
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_tmp -o pmc -- python tools/lde_only.py 2 > /dev/null 2>&1
  f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1); cp "$f" gpurun_out/r03n_pmc_$v.csv
  rm -rf gpurun_out/pmc_tmp
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc_tmp -o pmc -- python tools/lde_only.py 2 > /dev/null 2>&1
  f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03n_pmc2_$v.csv
done
rm -rf gpurun_out/pmc_tmp
python - <<'PY'
import csv, collections
for v in ("g4", "r8"):
    for tag in ("pmc", "pmc2"):
        try:
            rows = list(csv.DictReader(open("gpurun_out/r03n_%s_%s.csv" % (tag, v))))
        except Exception as e:
            print(v, tag, "missing", e); continue
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "ntt_pass" in k:
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in per.items():
            print(v, k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in d.items()}, "launches", len(next(iter(d.values()))))
PY
