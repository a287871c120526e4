#!/bin/bash
# round 6, GPU job p: zero-aware first NTT group of the 8x LDE: parity (LDE / prover tests), then timing with and without it
set -u
TAG=r06p; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_goldilocks.py tests/test_gpu_plonky2.py -x -q -m gpu > gpurun_out/${TAG}_pytest_goldilocks_plonky2.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/${TAG}_pytest_goldilocks_plonky2.log
( echo "== zero-aware first group (default)"; timeout 300 python tools/gl_quickbench.py 2>&1 | grep -v amdgpu.ids
  echo "== ZKLC_NTT_ZSKIP=0"; ZKLC_NTT_ZSKIP=0 timeout 300 python tools/gl_quickbench.py 2>&1 | grep -v amdgpu.ids
  echo "== Ed25519-shape proofs, default"; timeout 300 python tools/prove_profile_ed25519.py 6 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== Ed25519-shape proofs, ZKLC_NTT_ZSKIP=0"; ZKLC_NTT_ZSKIP=0 timeout 300 python tools/prove_profile_ed25519.py 6 2>&1 | grep -v amdgpu.ids | tail -2 ) > gpurun_out/${TAG}_lde_zero_aware_ab.txt 2>&1
cat gpurun_out/${TAG}_lde_zero_aware_ab.txt | grep -v "Merkle\|Poseidon\|iNTT"
