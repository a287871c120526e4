#!/bin/bash
# functional run of the strong-scaling mode: two ranks share the box's one GPU (gloo transport)
set -u
TAG=${1:-r02h}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --scaling strong --backend gloo --no-cpu-baseline --no-bn254-extras --msm-log 16 --witness-batch 16 > gpurun_out/${TAG}_strong2.json 2> gpurun_out/${TAG}_strong2.err; echo "rc=$?"
tail -c 1500 gpurun_out/${TAG}_strong2.err
python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/${TAG}_strong2.json").read().strip().split("\n")[-1])
    print({k:j[k] for k in ("metric","value","n_gpus","ms_per_step","scaling","final_proof_verified") if k in j})
    b=j["block_i"]; print({k:b[k] for k in b if k not in ("metric","note")})
except Exception as e: print("no json", e)
PY
