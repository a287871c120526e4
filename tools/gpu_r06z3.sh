#!/bin/bash
set -u
TAG=r06z3; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_groth16.py -x -q -m gpu 2>&1 | tail -2
( echo "== K on stream 1, Z after computeH on stream 3 (plain form)"; MODES=0 timeout 300 python tools/groth16_quickbench.py 22 5 2>&1 | grep "groth16 prove\|Error\|error" ) | tee gpurun_out/${TAG}_groth16_k_z_split.txt
