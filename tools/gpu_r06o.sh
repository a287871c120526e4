#!/bin/bash
# round 6, GPU job o: Groth16 prover on three streams + fixed-base tables (tests, then the 2^22 timing in both forms)
set -u
TAG=r06o; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_bn254.py -x -q -m gpu > gpurun_out/${TAG}_pytest_groth16_bn254.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/${TAG}_pytest_groth16_bn254.log
timeout 600 python tools/groth16_quickbench.py 22 4 > gpurun_out/${TAG}_groth16_quickbench.txt 2>&1; echo "quickbench rc=$?"
cat gpurun_out/${TAG}_groth16_quickbench.txt | tail -5
