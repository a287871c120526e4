#!/bin/bash
# round 6, GPU job f: the software-pipelined prove_stream (header thread, producer lookahead, per-thread chaining)
set -u
TAG=r06f; mkdir -p gpurun_out; export TMPDIR=/tmp
uptime > gpurun_out/${TAG}_host.txt; nproc >> gpurun_out/${TAG}_host.txt
timeout 1200 python -m pytest tests/test_gpu_stream_pipeline.py -x -q > gpurun_out/${TAG}_pytest_pipeline.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest_pipeline.log
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_C_detail.json timeout 900 $B > gpurun_out/${TAG}_C_poll_line.json 2> gpurun_out/${TAG}_C.err; echo "C rc=$?"; tail -3 gpurun_out/${TAG}_C.err
ZKLC_WAIT=event ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_B_detail.json timeout 900 $B > gpurun_out/${TAG}_B_event_line.json 2> gpurun_out/${TAG}_B.err; echo "B rc=$?"
ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_C2_detail.json timeout 900 $B > gpurun_out/${TAG}_C2_poll_line.json 2> gpurun_out/${TAG}_C2.err; echo "C2 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06f_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']
    print(f.split('r06f_')[1][:2], 's/block %.3f'%b['seconds_per_block'], 'per_step', b['per_step_s'], 'lat', b['latency_s'][:4], 'cores %.2f'%b['host_cores_busy'], 'rss %d'%b['rss_mb_after'], 'first %.1f'%b['first_block_s_incl_circuit_construction'], 'busy', b['telemetry_mean'].get('busy_pct'), 'W', b['telemetry_mean'].get('power_w'), 'fold', b['fold_thread_seconds'], 'dag', {k:round(v,2) for k,v in b['dag_thread_seconds'].items()}, b.get('host_load_before'))
PY
