#!/bin/bash
# round 6, GPU job v: settled read-backs (wait, THEN enqueue the device -> host copy) against copies parked behind their kernels, one box;
# prover byte parity first
set -u
TAG=r06v; mkdir -p gpurun_out; export TMPDIR=/tmp
uptime > gpurun_out/${TAG}_host.txt
timeout 900 python -m pytest tests/test_gpu_plonky2.py tests/test_gpu_witness.py tests/test_gpu_stream_pipeline.py -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${TAG}_pytest.log
B="python bench.py --steps 12 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
run() { name=$1; shift; env ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_${name}_detail.json "$@" > gpurun_out/${TAG}_${name}_line.json 2> gpurun_out/${TAG}_${name}.err; echo "$name rc=$?"; }
run a_settled1 timeout 900 $B
run b_parked1 ZKLC_SETTLED_COPIES=0 timeout 900 $B
run c_settled2 timeout 900 $B
run d_parked2 ZKLC_SETTLED_COPIES=0 timeout 900 $B
run e_settled3 timeout 900 $B
python - <<'PY' | tee gpurun_out/r06v_settled_copies_ab.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r06v_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']; ps=b['per_step_s']
    pm=d['stages']['prove'] if 'prove' in d.get('stages',{}) else {}
    ed=[v.get('ms_per_proof') for k,v in pm.items() if isinstance(v,dict) and k.startswith('ed25519_circuit')]
    print('%-12s'%f.split('r06v_')[1].split('_detail')[0], 's/block %.3f'%b['seconds_per_block'], 'mid %.3f'%(sum(ps[2:-1])/len(ps[2:-1])), 'first %.2f last %.2f'%(ps[0],ps[-1]), 'busy', b['telemetry_mean'].get('busy_pct'), 'W', b['telemetry_mean'].get('power_w'), 'cores %.2f'%b['host_cores_busy'], 'ed25519 proof ms', ed[:1], 'wit s', round(b['witness_producer_seconds'],2))
PY
