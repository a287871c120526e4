#!/bin/bash
# round 6, GPU job g: hardware queues -- HIP maps the normal-priority streams of a process onto GPU_MAX_HW_QUEUES (default 4) queues;
# the pipeline has six (torch's, three Ed25519 provers, witness producer, keys / stakes) + three high-priority ones
set -u
TAG=r06g; mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras --no-stages-msm"
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
run() { # name, env..., -- extra args
  name=$1; shift
  env ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_${name}_detail.json "$@" timeout 900 $B $EXTRA > gpurun_out/${TAG}_${name}_line.json 2> gpurun_out/${TAG}_${name}.err; echo "$name rc=$?"
}
EXTRA=""; run q4 GPU_MAX_HW_QUEUES=4
EXTRA=""; run q8 GPU_MAX_HW_QUEUES=8
EXTRA="--prove-streams 5"; run q8s5 GPU_MAX_HW_QUEUES=8
EXTRA=""; run q12 GPU_MAX_HW_QUEUES=12
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06g_*_detail.json')):
    d=json.load(open(f)); b=d['block_i']
    ps=b['per_step_s']
    print(f.split('r06g_')[1].split('_detail')[0], 's/block %.3f'%b['seconds_per_block'], 'steady %.3f'%((sum(ps)-ps[0])/(len(ps)-1)), 'per_step', ps, 'lat', b['latency_s'][:3], 'cores %.2f'%b['host_cores_busy'], 'busy', b['telemetry_mean'].get('busy_pct'), 'W', b['telemetry_mean'].get('power_w'), 'ed ms %.1f'%d['stages']['prove']['ed25519_circuit_2p18x234']['ms_per_proof'])
PY
