#!/bin/bash
# round 6, GPU job i: which of the three prove_stream changes pay, on ONE box (ZKLC_STREAM), 12 timed blocks each
set -u
TAG=r06i; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
run() { name=$1; shift; env ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_${name}_detail.json "$@" timeout 900 $B > gpurun_out/${TAG}_${name}_line.json 2> gpurun_out/${TAG}_${name}.err; echo "$name rc=$?"; }
run warm ZKLC_STREAM=stages
run stages ZKLC_STREAM=stages
run fold ZKLC_STREAM=fold
run fold_hdr ZKLC_STREAM=fold,hdr
run all ZKLC_STREAM=fold,hdr,ahead
run stages_q4 ZKLC_STREAM=stages GPU_MAX_HW_QUEUES=4
run all2 ZKLC_STREAM=fold,hdr,ahead
run stages2 ZKLC_STREAM=stages
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06i_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']; ps=b['per_step_s']
    print('%-10s'%f.split('r06i_')[1].split('_detail')[0], 's/block %.3f'%b['seconds_per_block'], 'mid %.3f'%(sum(ps[2:-1])/len(ps[2:-1])), 'first %.2f last %.2f'%(ps[0],ps[-1]), 'lat %.1f'%(sum(b['latency_s'])/len(ps)), 'cores %.2f'%b['host_cores_busy'], 'busy', b['telemetry_mean'].get('busy_pct'), 'W', b['telemetry_mean'].get('power_w'))
PY
