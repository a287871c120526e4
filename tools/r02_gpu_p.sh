#!/bin/bash
# round 2, final validation: the full GPU suite, smoke(), the default bench line, the rocprofv3 summary of the bench command
set -u
TAG=${1:-r02p}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -18 gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -c 400 gpurun_out/${TAG}_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().split("\n")[-1])
print({k:j[k] for k in ("metric","value","unit","ms_per_step","final_proof_verified","n_gpus","steps","warmup") if k in j})
print("roofline", j["roofline"])
print("cpu_baseline", j.get("cpu_baseline"))
b=j["block_i"]
print({k:b[k] for k in ("seconds_per_block","seconds_until_signature_aggregate","fold_thread_seconds","dag_thread_seconds")})
print(j["stages"]["prove"]["ed25519_circuit_2p18x234"]["stages_ms"])
print({k: (v.get("ms"), v.get("value")) for k, v in j["stages"].items() if isinstance(v, dict) and "ms" in v})
PY
rm -rf gpurun_out/prof_tmp
timeout 480 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o bench -- python bench.py --steps 1 --warmup 1 --no-bn254-extras --no-cpu-baseline > gpurun_out/${TAG}_bench_rocprof.json 2> gpurun_out/${TAG}_bench_rocprof.err; echo "rocprof bench rc=$?"
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_kernel_stats.csv && head -12 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-70,110-230
rm -rf gpurun_out/prof_tmp
