#!/bin/bash
# round 3, GPU call r: the default multi-rank bench line, two ranks sharing the one GPU of the test box over gloo (functional check of
# the weak region with overlapped blocks, the strong block section and the sharded MSM section)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --backend gloo --no-cpu-baseline --no-bn254-extras > gpurun_out/r03r_bench_2ranks_gloo.json 2> gpurun_out/r03r_bench_2ranks_gloo.err
echo "rc=$?"; tail -c 600 gpurun_out/r03r_bench_2ranks_gloo.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r03r_bench_2ranks_gloo.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("metric", "value", "n_gpus", "ms_per_step", "scaling", "final_proof_verified")})
print("strong block:", j["block_i"].get("strong"))
print("strong msm:", j["stages"]["msm"].get("strong"))
PY
