#!/bin/bash
# Generic PMC passes for the kernels of one command:  tools/pmc_kernels.sh TAG KERNEL_PREFIX LAUNCHES -- <command ...>
# Three rocprofv3 runs (--pmc only with --kernel-trace; never combined with another trace domain): SQ counters, FETCH_SIZE,
# WRITE_SIZE.  Writes gpurun_out/TAG_pmc.json: per kernel whose name starts with KERNEL_PREFIX the per-launch means, and the totals per
# pipeline invocation (the command runs the pipeline LAUNCHES times).  traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; the x2 is
# the gfx950 correction for wide streaming reads of MI355X_MICROARCH.md -- gather kernels are reported with both readings).
set -u
TAG=$1; PREFIX=$2; LAUNCHES=$3; shift 4
export TMPDIR=/tmp
mkdir -p gpurun_out
for pass in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf gpurun_out/pmc_tmp
  timeout 900 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -o pmc -- "$@" > gpurun_out/${TAG}_pmc.log 2>&1
  f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1)
  cp "$f" "gpurun_out/${TAG}_pmc_$(echo $pass | cut -d' ' -f1).csv"
done
rm -rf gpurun_out/pmc_tmp
python - "$TAG" "$PREFIX" "$LAUNCHES" <<'PY'
import csv, json, sys, collections
tag, prefix, launches = sys.argv[1], sys.argv[2], int(sys.argv[3])
out = {"kernels": {}, "pipeline_invocations": launches}
for name in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open("gpurun_out/%s_pmc_%s.csv" % (tag, name))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(prefix):
            per[k][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, disp in per.items():
        d = out["kernels"].setdefault(k, {})
        for cname in sorted(set(c for v in disp.values() for c in v)):
            vals = [v[cname] for v in disp.values() if cname in v]
            d[cname + "_per_launch"] = sum(vals) / len(vals)
        d["launches_per_invocation"] = len(disp) / launches
tot = lambda key, scale=1.0: sum(d.get(key, 0) * scale * d["launches_per_invocation"] for d in out["kernels"].values())
out["valu_wave_instructions_per_invocation"] = tot("SQ_INSTS_VALU_per_launch")
out["fetch_bytes_raw_per_invocation"] = tot("FETCH_SIZE_per_launch", 1024)
out["write_bytes_per_invocation"] = tot("WRITE_SIZE_per_launch", 1024)
out["hbm_bytes_per_invocation"] = 2 * out["fetch_bytes_raw_per_invocation"] + out["write_bytes_per_invocation"]
out["note"] = "traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes); gathers of sub-line records count once (see fetch_bytes_raw)"
json.dump(out, open("gpurun_out/%s_pmc.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
PY
rm -f gpurun_out/${TAG}_pmc_SQ_INSTS_VALU.csv gpurun_out/${TAG}_pmc_FETCH_SIZE.csv gpurun_out/${TAG}_pmc_WRITE_SIZE.csv
