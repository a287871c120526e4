#!/bin/bash
# round 3, GPU call u: ZKLC_CIRCUIT_CACHE on an empty directory (entries written), then again (entries loaded: the proofs must verify)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export ZKLC_CIRCUIT_CACHE=/tmp/zklc_cache
rm -rf $ZKLC_CIRCUIT_CACHE
for pass in write read; do
  t0=$SECONDS
  timeout 900 python -m pytest tests/test_gpu_plonky2.py tests/test_gpu_sha256.py -x -q -k "mainnet_signature or sha256" > gpurun_out/r03u_pytest_cache_$pass.log 2>&1; echo "cache $pass rc=$? in $((SECONDS - t0)) s"; tail -2 gpurun_out/r03u_pytest_cache_$pass.log
  ls -la $ZKLC_CIRCUIT_CACHE | head -12
done
