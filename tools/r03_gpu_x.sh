#!/bin/bash
# round 3, GPU call x: proofs in flight per GPU: 6 streams against the default 4
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 5 --warmup 2 --prove-streams 6 --no-cpu-baseline --no-bn254-extras > gpurun_out/r03x_bench_streams6.json 2> gpurun_out/r03x_bench_streams6.err; echo "rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03x_bench_streams6.json").read().strip().splitlines()[-1])
b = j["block_i"]
print(j["value"], j["ms_per_step"], b["per_step_s"], b["streams"], [t["busy_pct"] for t in b["per_step_telemetry"]])
PY
