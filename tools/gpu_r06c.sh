#!/bin/bash
# round 6, third GPU job: the scalar-unit Poseidon reduction (parity + timing) and the wait modes A/B/C on one box
set -u
TAG=r06c; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_goldilocks.py tests/test_gpu_plonky2.py tests/test_gpu_witness.py -x -q > gpurun_out/${TAG}_pytest_poseidon.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest_poseidon.log
for m in poll event spin; do
  ZKLC_WAIT=$m timeout 300 python tools/host_cpu_probe.py 17 10 2>/dev/null | sed "s/^/[wait=$m] /" >> gpurun_out/${TAG}_host_cpu_probe.txt
done
cat gpurun_out/${TAG}_host_cpu_probe.txt
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
ZKLC_WAIT=event ZKLC_PINNED_STAGING=0 ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_A_detail.json timeout 900 $B > gpurun_out/${TAG}_A_event_pageable_line.json 2> gpurun_out/${TAG}_A.err; echo "A rc=$?"
ZKLC_WAIT=event ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_B_detail.json timeout 900 $B > gpurun_out/${TAG}_B_event_pinned_line.json 2> gpurun_out/${TAG}_B.err; echo "B rc=$?"
ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_C_detail.json timeout 900 $B > gpurun_out/${TAG}_C_poll_pinned_line.json 2> gpurun_out/${TAG}_C.err; echo "C rc=$?"
ZKLC_WAIT=event ZKLC_PINNED_STAGING=0 ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_A2_detail.json timeout 900 $B > gpurun_out/${TAG}_A2_event_pageable_line.json 2> gpurun_out/${TAG}_A2.err; echo "A2 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06c_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']
    print(f.split('r06c_')[1][:2], 's/block %.3f'%b['seconds_per_block'], 'per_step', b['per_step_s'], 'cores %.2f'%b['host_cores_busy'], 'rss %d'%b['rss_mb_after'], 'first %.1f'%b['first_block_s_incl_circuit_construction'], 'merkle ms %.2f'%d['stages']['merkle']['ms'], 'ed ms', d['stages']['prove']['ed25519_circuit_2p18x234']['ms_per_proof'])
PY
