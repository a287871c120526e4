#!/bin/bash
# round 6, GPU job j: proofs in flight (--prove-streams = Ed25519 provers + 1) under the software pipeline with 8 hardware queues
set -u
TAG=r06j; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
run() { name=$1; shift; ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_${name}_detail.json timeout 900 $B "$@" > gpurun_out/${TAG}_${name}_line.json 2> gpurun_out/${TAG}_${name}.err; echo "$name rc=$?"; }
run warm
run s4
run s3 --prove-streams 3
run s4b
run s3b --prove-streams 3
run s2 --prove-streams 2
run s4w16 --witness-batch 16
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06j_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']; ps=b['per_step_s']
    print('%-8s'%f.split('r06j_')[1].split('_detail')[0], 's/block %.3f'%b['seconds_per_block'], 'mid %.3f'%(sum(ps[2:-1])/len(ps[2:-1])), 'first %.2f last %.2f'%(ps[0],ps[-1]), 'lat %.1f'%(sum(b['latency_s'])/len(ps)), 'cores %.2f'%b['host_cores_busy'], 'busy', b['telemetry_mean'].get('busy_pct'), 'W', b['telemetry_mean'].get('power_w'))
PY
