#!/bin/bash
# round 3, GPU call t: the circuit-construction paths that now go through plonky2/circuit_cache.py -- without the cache, then with
# ZKLC_CIRCUIT_CACHE on an empty directory (entries written) and again (entries loaded: the proofs must still verify)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_plonky2.py tests/test_gpu_sha256.py -x -q -k "approvals or mainnet_signature or tree or sha256" > gpurun_out/r03t_pytest_nocache.log 2>&1; echo "no cache rc=$?"; tail -2 gpurun_out/r03t_pytest_nocache.log
export ZKLC_CIRCUIT_CACHE=/tmp/zklc_cache
rm -rf $ZKLC_CIRCUIT_CACHE
for pass in write read; do
  /usr/bin/time -f "$pass pass: %e s" timeout 900 python -m pytest tests/test_gpu_plonky2.py tests/test_gpu_sha256.py -x -q -k "mainnet_signature or sha256" > gpurun_out/r03t_pytest_cache_$pass.log 2>&1; echo "cache $pass rc=$?"; tail -2 gpurun_out/r03t_pytest_cache_$pass.log
  ls -la $ZKLC_CIRCUIT_CACHE | head -12
done
