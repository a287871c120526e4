#!/bin/bash
# PMC passes of the dominant kernel of a block proof (gl_hash_leaves_kernel at 2^20 x 234): SQ counters, FETCH_SIZE, WRITE_SIZE in
# separate rocprofv3 runs (the TCC counters do not fit one pass); writes profiles-ready JSON gpurun_out/<tag>_poseidon_pmc.json
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
declare -A RES
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf gpurun_out/pmc_tmp
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d gpurun_out/pmc_tmp -o pmc -- python tools/merkle_only.py 3 > gpurun_out/${TAG}_pmc_merkle.log 2>&1
  f=$(find gpurun_out/pmc_tmp -name '*counter_collection.csv' | head -1)
  cp "$f" "gpurun_out/${TAG}_pmc_merkle_$(echo $pass | cut -d' ' -f1).csv"
done
rm -rf gpurun_out/pmc_tmp
python - "$TAG" <<'PY'
import csv, json, sys, collections
tag = sys.argv[1]
out = {"kernel": "gl_hash_leaves_kernel", "leaves": 1 << 20, "width": 234}
for name in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
    rows = [r for r in csv.DictReader(open("gpurun_out/%s_pmc_merkle_%s.csv" % (tag, name))) if r["Kernel_Name"].startswith("gl_hash_leaves_kernel")]
    per = collections.defaultdict(dict)
    for r in rows:
        per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for cname in sorted(set(c for d in per.values() for c in d)):
        vals = [d[cname] for d in per.values() if cname in d]
        out[cname + "_per_launch"] = sum(vals) / len(vals)
    out["launches"] = len(per)
# FETCH_SIZE / WRITE_SIZE are reported in KiB
if "FETCH_SIZE_per_launch" in out:
    out["hbm_bytes_per_launch"] = 2 * out["FETCH_SIZE_per_launch"] * 1024 + out.get("WRITE_SIZE_per_launch", 0) * 1024
    out["note"] = "traffic = 2 x FETCH_SIZE (gfx950 correction for wide streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE"
json.dump(out, open("gpurun_out/%s_poseidon_pmc.json" % tag, "w"), indent=1)
print(json.dumps(out))
PY
