#!/bin/bash
# round 6, GPU job y: the host-pointer entry points with settled read-backs (tests of every module that has one), the C-ABI caller,
# and a short bench on the final library
set -u
TAG=r06y; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bn254.py tests/test_gpu_ed25519.py tests/test_gpu_goldilocks.py tests/test_gpu_groth16.py tests/test_gpu_c_abi.py tests/test_gpu_epoch.py tests/test_gpu_sha256.py -x -q -m gpu > gpurun_out/${TAG}_pytest_host_entry_points.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${TAG}_pytest_host_entry_points.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${TAG}_smoke.log
ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_bench_detail.json timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --c5-validators 0 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
l=json.load(open('gpurun_out/r06y_bench_line.json')); b=l['block_i']
print('s/block',b['seconds_per_block'],'steps',b['per_step_s'],'busy',b['gpu']['busy_pct'],'groth16',l['stages']['groth16_prove_2p22'],'msm',l['stages']['msm']['ms'])
PY
