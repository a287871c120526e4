#!/bin/bash
# round 2, call o: recursion witnesses on the GPU (fold chain + DAG) -- tests, then the block bench
set -u
TAG=${1:-r02o}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_recursion.py tests/test_gpu_sha256.py tests/test_gpu_witness.py -m gpu -x -q -k "not reference_ed25519" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest.log
timeout 1500 python bench.py --steps 2 --warmup 1 --no-bn254-extras --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/${TAG}_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().split("\n")[-1])
print({k:j[k] for k in ("metric","value","ms_per_step","final_proof_verified") if k in j})
b=j["block_i"]
print({k:b[k] for k in b if k not in ("metric","note","dag_thread_counts")})
PY
