#!/bin/bash
set -u
TAG=${1:-r02i}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_epoch.py -m gpu -x -q -s > gpurun_out/${TAG}_epoch.log 2>&1; echo "epoch rc=$?"; tail -5 gpurun_out/${TAG}_epoch.log
bash tools/pmc_merkle.sh ${TAG} 2>&1 | tail -3
timeout 1500 python bench.py --steps 2 --warmup 1 --c5-validators 200 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -c 1200 gpurun_out/${TAG}_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().split("\n")[-1])
print({k:j[k] for k in ("metric","value","ms_per_step","final_proof_verified") if k in j})
print("roofline", j["roofline"])
print("c5", j["stages"]["prove"].get("c5_synthetic_epoch"))
print("extras", j["stages"].get("bn254_extras"))
print("cpu_baseline", j.get("cpu_baseline"))
PY
