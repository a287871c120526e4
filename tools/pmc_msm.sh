#!/bin/bash
# PMC passes of the BN254 G1 MSM at 2^22 (tools/msm_quickbench.py 22): SQ counters, FETCH_SIZE, WRITE_SIZE in separate rocprofv3
# runs (--pmc only, no other trace domain); writes gpurun_out/<tag>_msm_pmc.json (copy to profiles/msm_pmc_latest.json: bench.py
# reads it for stages.msm.roofline.traffic / .valu)
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf gpurun_out/pmc_tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -o pmc -- python tools/msm_quickbench.py 22 > gpurun_out/${TAG}_pmc_msm.log 2>&1
  f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1)
  cp "$f" "gpurun_out/${TAG}_pmc_msm_$(echo $pass | cut -d' ' -f1).csv"
done
rm -rf gpurun_out/pmc_tmp
python - "$TAG" <<'PY'
import csv, json, sys, collections
tag = sys.argv[1]
out = {"log_n": 22, "kernels": {}}
for name in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open("gpurun_out/%s_pmc_msm_%s.csv" % (tag, name))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("msm_"):
            per[k][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, disp in per.items():
        d = out["kernels"].setdefault(k, {})
        for cname in sorted(set(c for v in disp.values() for c in v)):
            vals = [v[cname] for v in disp.values() if cname in v]
            d[cname + "_per_launch"] = sum(vals) / len(vals)
        d["launches_per_msm"] = max(1, round(len(disp) / 4))      # the quick bench runs 4 multi-exponentiations
GATHER = ("msm_slice_kernel",)      # 80-byte point records straddle two 64-byte lines: 16 windows x 2^22 x 128 B = 8.6 GB predicted,
                                    # the raw counter reads 8.7 GB -> the x2 of wide streaming reads does not apply to this kernel
def fetch_bytes(k, d):
    return d.get("FETCH_SIZE_per_launch", 0) * 1024 * (1 if k.startswith(GATHER) else 2)
for k, d in out["kernels"].items():
    d["hbm_bytes_per_launch"] = fetch_bytes(k, d) + d.get("WRITE_SIZE_per_launch", 0) * 1024
out["hbm_bytes_per_msm"] = sum(d["hbm_bytes_per_launch"] * d["launches_per_msm"] for d in out["kernels"].values())
out["hbm_bytes_per_msm_raw_counters"] = sum((d.get("FETCH_SIZE_per_launch", 0) + d.get("WRITE_SIZE_per_launch", 0)) * 1024 * d["launches_per_msm"]
                                            for d in out["kernels"].values())
out["valu_wave_instructions_per_msm"] = sum(d.get("SQ_INSTS_VALU_per_launch", 0) * d["launches_per_msm"] for d in out["kernels"].values())
out["note"] = ("traffic = FETCH_SIZE (x2 for the streaming kernels: gfx950 correction for wide reads, MI355X_MICROARCH.md; x1 for the "
               "gathers of msm_slice_kernel, calibrated on its known 128 B per gathered point) + WRITE_SIZE, KiB -> bytes, all kernels of "
               "one 2^22 multi-exponentiation")
json.dump(out, open("gpurun_out/%s_msm_pmc.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
PY
