#!/bin/bash
# round 3, GPU call m: Goldilocks NTT passes with shift twiddles (groups of 4 stages): parity + A/B timing against the radix-8 kernel
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_goldilocks.py -x -q > gpurun_out/r03m_pytest_gl.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03m_pytest_gl.log
timeout 300 python tools/ntt_quickbench.py > gpurun_out/r03m_ntt_g4.txt 2>&1; cat gpurun_out/r03m_ntt_g4.txt
ZKLC_NTT_R8=1 timeout 300 python tools/ntt_quickbench.py > gpurun_out/r03m_ntt_r8.txt 2>&1; sed 's/^/r8: /' gpurun_out/r03m_ntt_r8.txt
