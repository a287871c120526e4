#!/bin/bash
# round 2, call n: block-level bench after the kernel changes of calls j..m
set -u
TAG=${1:-r02n}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python bench.py --steps 2 --warmup 1 --no-bn254-extras --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/${TAG}_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().split("\n")[-1])
print({k:j[k] for k in ("metric","value","ms_per_step","final_proof_verified") if k in j})
b=j["block_i"]
print({k:b[k] for k in b if k not in ("metric","note","dag_thread_counts")})
print(j["stages"]["prove"]["ed25519_circuit_2p18x234"]["stages_ms"])
print("merkle", j["stages"]["merkle"]["ms"], "roofline", j["roofline"])
PY
