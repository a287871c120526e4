#!/bin/bash
# round 6, GPU job l: witness buffers 3 / 4, with and without the small first chunk, one box
set -u
TAG=r06l; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
run() { name=$1; shift; env ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_${name}_detail.json "$@" > gpurun_out/${TAG}_${name}_line.json 2> gpurun_out/${TAG}_${name}.err; echo "$name rc=$?"; }
run warm timeout 900 $B
run b3 timeout 900 $B
run b3small ZKLC_WIT_SMALL_FIRST=always timeout 900 $B
run b4 ZKLC_WIT_BUFS=4 timeout 900 $B
run b3b timeout 900 $B
run b2 ZKLC_WIT_BUFS=2 timeout 900 $B
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06l_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']; ps=b['per_step_s']
    print('%-8s'%f.split('r06l_')[1].split('_detail')[0], 's/block %.3f'%b['seconds_per_block'], 'mid %.3f'%(sum(ps[2:-1])/len(ps[2:-1])), 'first %.2f last %.2f'%(ps[0],ps[-1]), 'lat %.1f'%(sum(b['latency_s'])/len(ps)), 'wit %.2f'%b['witness_producer_seconds'], 'cores %.2f'%b['host_cores_busy'], 'busy', b['telemetry_mean'].get('busy_pct'), 'W', b['telemetry_mean'].get('power_w'))
PY
