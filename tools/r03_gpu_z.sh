#!/bin/bash
# round 3, final validation: the full GPU suite, smoke(), the DRIVER's bench command, the PMC passes of the LDE, and the rocprofv3
# summary of a short bench run
set -u
TAG=${1:-r03z}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -18 gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2> gpurun_out/${TAG}_bench_driver_cmd.err; echo "bench rc=$?"
tail -c 300 gpurun_out/${TAG}_bench_driver_cmd.err
python - <<PY
import json
j=json.loads(open("gpurun_out/${TAG}_bench_driver_cmd.json").read().strip().split("\n")[-1])
print({k:j[k] for k in ("metric","value","unit","ms_per_step","final_proof_verified","n_gpus","steps","warmup") if k in j})
print("roofline", j["roofline"])
print("cpu_baseline", j.get("cpu_baseline"))
b=j["block_i"]
print({k:b[k] for k in ("seconds_per_block","per_step_s","fold_thread_seconds","first_block_s_incl_circuit_construction")})
print(j["stages"]["prove"]["ed25519_circuit_2p18x234"].get("stages_ms"), j["stages"]["prove"]["ed25519_circuit_2p18x234"]["ms_per_proof"])
print({k: (v.get("ms"), v.get("value")) for k, v in j["stages"].items() if isinstance(v, dict) and "ms" in v})
PY
bash tools/pmc_lde.sh ${TAG}
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o bench -- python bench.py --steps 1 --warmup 1 --no-bn254-extras --no-cpu-baseline > gpurun_out/${TAG}_bench_rocprof.json 2> gpurun_out/${TAG}_bench_rocprof.err; echo "rocprof bench rc=$?"
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_bench_kernel_stats.csv && head -12 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-70,110-230
rm -rf gpurun_out/prof_tmp
