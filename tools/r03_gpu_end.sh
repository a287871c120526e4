#!/bin/bash
# round 3, end: the default bench line on the final tree (after the NTT load batching), short form
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 235 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bn254-extras > gpurun_out/r03_end_bench.json 2> gpurun_out/r03_end_bench.err; echo "rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03_end_bench.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["final_proof_verified"], j["block_i"]["per_step_s"], j["stages"]["prove"]["ed25519_circuit_2p18x234"]["ms_per_proof"], j["stages"]["lde"]["ms"], j["stages"]["lde"]["roofline"]["valu"]["frac"])
PY
