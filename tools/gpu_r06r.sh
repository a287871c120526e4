#!/bin/bash
# round 6, GPU job r: what stops the Python threads of the pipeline?  collector pauses + interpreter stalls measured in the timed region;
# automatic collections against one young-generation collection per block, alternating on one box
set -u
TAG=r06r; mkdir -p gpurun_out; export TMPDIR=/tmp
uptime > gpurun_out/${TAG}_host.txt
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
run() { name=$1; shift; env ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_${name}_detail.json "$@" > gpurun_out/${TAG}_${name}_line.json 2> gpurun_out/${TAG}_${name}.err; echo "$name rc=$?"; }
run warm timeout 900 $B
run auto1 timeout 900 $B
run block1 ZKLC_BENCH_GC=block timeout 900 $B
run auto2 timeout 900 $B
run block2 ZKLC_BENCH_GC=block timeout 900 $B
python - <<'PY' | tee gpurun_out/r06r_gc_ab.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r06r_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']; ps=b['per_step_s']
    print('%-8s'%f.split('r06r_')[1].split('_detail')[0], 's/block %.3f'%b['seconds_per_block'], 'mid %.3f'%(sum(ps[2:-1])/len(ps[2:-1])), 'first %.2f last %.2f'%(ps[0],ps[-1]), 'busy', b['telemetry_mean'].get('busy_pct'), 'cores %.2f'%b['host_cores_busy'], 'rss %.0f'%b['rss_mb_after'], 'gc', b['gc'], 'stalls', {k:v for k,v in b['interpreter_stalls'].items() if k!='note'})
PY
