#!/bin/bash
# round 3, GPU call s: the sharded-MSM strong section with two ranks on one GPU (gloo), no proving stage
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 1 --warmup 1 --backend gloo --no-prove --no-cpu-baseline --no-bn254-extras > gpurun_out/r03s_bench_2ranks_msm.json 2> gpurun_out/r03s_bench_2ranks_msm.err
echo "rc=$?"; tail -c 400 gpurun_out/r03s_bench_2ranks_msm.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03s_bench_2ranks_msm.json").read().strip().splitlines()[-1])
print(j["stages"]["msm"])
PY
timeout 600 python -m pytest tests/test_gpu_groth16.py -x -q 2>&1 | tail -3
