#!/bin/bash
set -u
TAG=r06z2; mkdir -p gpurun_out; export TMPDIR=/tmp
( echo "== stream 3 high priority"; MODES=0 timeout 300 python tools/groth16_quickbench.py 22 5 2>&1 | grep "groth16 prove"
  echo "== ZKLC_GROTH16_H_PRIORITY=0"; ZKLC_GROTH16_H_PRIORITY=0 MODES=0 timeout 300 python tools/groth16_quickbench.py 22 5 2>&1 | grep "groth16 prove"
  echo "== stream 3 high priority (again)"; MODES=0 timeout 300 python tools/groth16_quickbench.py 22 5 2>&1 | grep "groth16 prove" ) | tee gpurun_out/${TAG}_groth16_h_priority_ab.txt
timeout 300 python -m pytest tests/test_gpu_groth16.py -x -q -m gpu 2>&1 | tail -1
