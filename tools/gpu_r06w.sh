#!/bin/bash
# round 6, GPU job w: with the read-backs settled the witness producer runs 4.3 s of a 4.6 s block: producer priority, buffers, prover streams
set -u
TAG=r06w; mkdir -p gpurun_out; export TMPDIR=/tmp
uptime > gpurun_out/${TAG}_host.txt
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
run() { name=$1; shift; env ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_${name}_detail.json "$@" > gpurun_out/${TAG}_${name}_line.json 2> gpurun_out/${TAG}_${name}.err; echo "$name rc=$?"; }
run a_default timeout 900 $B
run b_witprio ZKLC_WIT_PRIORITY=1 timeout 900 $B
run c_streams5 timeout 900 $B --prove-streams 5
run d_bufs4 ZKLC_WIT_BUFS=4 timeout 900 $B
run e_bufs2 ZKLC_WIT_BUFS=2 timeout 900 $B
run f_default2 timeout 900 $B
run g_batch16 timeout 900 $B --witness-batch 16
python - <<'PY' | tee gpurun_out/r06w_pipeline_variants_after_settled_copies.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r06w_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']; ps=b['per_step_s']
    print('%-12s'%f.split('r06w_')[1].split('_detail')[0], 's/block %.3f'%b['seconds_per_block'], 'mid %.3f'%(sum(ps[2:-1])/len(ps[2:-1])), 'first %.2f last %.2f'%(ps[0],ps[-1]), 'busy', b['telemetry_mean'].get('busy_pct'), 'W', b['telemetry_mean'].get('power_w'), 'cores %.2f'%b['host_cores_busy'], 'wit s', round(b['witness_producer_seconds'],2), 'hbm', b.get('hbm_used_gb'))
PY
