#!/bin/bash
# round 6, GPU job s: the interpreter's thread switch interval under prove_stream (per-block collections are the default now), one box
set -u
TAG=r06s; mkdir -p gpurun_out; export TMPDIR=/tmp
uptime > gpurun_out/${TAG}_host.txt
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
run() { name=$1; shift; env ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_${name}_detail.json "$@" > gpurun_out/${TAG}_${name}_line.json 2> gpurun_out/${TAG}_${name}.err; echo "$name rc=$?"; }
run a_warm timeout 900 $B
run b_def1 timeout 900 $B
run c_sw05_1 ZKLC_SWITCH_INTERVAL_MS=0.5 timeout 900 $B
run d_def2 timeout 900 $B
run e_sw05_2 ZKLC_SWITCH_INTERVAL_MS=0.5 timeout 900 $B
run f_sw02 ZKLC_SWITCH_INTERVAL_MS=0.2 timeout 900 $B
run g_auto ZKLC_STREAM_GC=auto timeout 900 $B
python - <<'PY' | tee gpurun_out/r06s_switch_interval_ab.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r06s_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']; ps=b['per_step_s']
    print('%-10s'%f.split('r06s_')[1].split('_detail')[0], 's/block %.3f'%b['seconds_per_block'], 'mid %.3f'%(sum(ps[2:-1])/len(ps[2:-1])), 'first %.2f last %.2f'%(ps[0],ps[-1]), 'busy', b['telemetry_mean'].get('busy_pct'), 'cores %.2f'%b['host_cores_busy'], 'gc', b['gc']['mode'], b['gc']['switch_interval_ms'], b['gc']['pause_s_per_block'], 'stalls', {k:v for k,v in b['interpreter_stalls'].items() if k!='note'})
PY
