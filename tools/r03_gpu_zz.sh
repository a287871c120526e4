#!/bin/bash
# round 3, last GPU call: batched tile loads in the NTT passes -- NTT parity, timing, prover byte parity at the reference sizes, smoke
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_goldilocks.py -x -q > gpurun_out/r03zz_pytest_gl.log 2>&1; echo "gl rc=$?"; tail -2 gpurun_out/r03zz_pytest_gl.log
timeout 200 python tools/ntt_quickbench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03zz_ntt.txt
timeout 600 python -m pytest tests/test_gpu_plonky2.py -x -q -k "bit_for_bit or c_prover or reference_sized" > gpurun_out/r03zz_pytest_p2.log 2>&1; echo "p2 rc=$?"; tail -2 gpurun_out/r03zz_pytest_p2.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
