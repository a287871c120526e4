#!/bin/bash
set -u
TAG=${1:-r02g}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_plonky2.py tests/test_gpu_witness.py -m gpu -x -q -k "approvals or mainnet or device_witness" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${TAG}_pytest.log
timeout 1200 python bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -c 800 gpurun_out/${TAG}_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().split("\n")[-1])
print({k:j[k] for k in ("metric","value","ms_per_step","final_proof_verified") if k in j})
b=j["block_i"]
print({k:b[k] for k in b if k not in ("metric","note","dag_thread_counts")})
print(j["stages"]["prove"]["ed25519_circuit_2p18x234"]["stages_ms"])
print("merkle", j["stages"]["merkle"]["ms"], "roofline", j["roofline"])
PY
