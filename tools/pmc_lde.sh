#!/bin/bash
# PMC passes of the coset LDE (tools/lde_only.py: 234 x 2^LOGN -> 2^(LOGN+3), LOGN = $2, default 17 = C3; 18 = the product shape): SQ counters, FETCH_SIZE, WRITE_SIZE in separate rocprofv3
# runs (--pmc only); writes gpurun_out/<tag>_lde_pmc.json (copy to profiles/lde_pmc_latest.json: bench.py reads it)
set -u
TAG=${1:-r03}
LOGN=${2:-17}
export TMPDIR=/tmp
for pass in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf gpurun_out/pmc_tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -o pmc -- python tools/lde_only.py 3 $LOGN > gpurun_out/${TAG}_pmc_lde.log 2>&1
  f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1)
  cp "$f" "gpurun_out/${TAG}_pmc_lde_$(echo $pass | cut -d' ' -f1).csv"
done
rm -rf gpurun_out/pmc_tmp
python - "$TAG" "$LOGN" <<'PY'
import csv, json, sys, collections
tag = sys.argv[1]
out = {"shape": "234 x 2^%d -> 2^%d" % (int(sys.argv[2]), int(sys.argv[2]) + 3), "kernels": {}}
for name in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open("gpurun_out/%s_pmc_lde_%s.csv" % (tag, name))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("gl_"):
            per[k][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, disp in per.items():
        d = out["kernels"].setdefault(k, {})
        for cname in sorted(set(c for v in disp.values() for c in v)):
            vals = [v[cname] for v in disp.values() if cname in v]
            d[cname + "_per_launch"] = sum(vals) / len(vals)
        d["launches_per_lde"] = max(1, round(len(disp) / 3))       # lde_only.py runs 3 extensions
tot = lambda key: sum(d.get(key, 0) * d["launches_per_lde"] for d in out["kernels"].values())
out["hbm_bytes_per_lde"] = 2 * tot("FETCH_SIZE_per_launch") * 1024 + tot("WRITE_SIZE_per_launch") * 1024
out["valu_wave_instructions_per_lde"] = tot("SQ_INSTS_VALU_per_launch")
out["note"] = "traffic = 2 x FETCH_SIZE (gfx950 correction for wide streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE, KiB -> bytes, all passes of one extension"
json.dump(out, open("gpurun_out/%s_lde_pmc.json" % tag, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
PY
