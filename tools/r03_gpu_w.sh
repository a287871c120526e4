#!/bin/bash
# round 3, GPU call w: the slow-gated parity checks on the final tree -- the real 2^18 x 234 Ed25519-circuit proof byte for byte against
# the oracle's C prover (and the Python generators' wire matrix cell for cell), default quotient path and ZKLC_P2_ADDMANY=multi
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
export ZKLC_SLOW_TESTS=1
timeout 1500 python -m pytest tests/test_gpu_plonky2.py -x -q -k "mainnet_signature" > gpurun_out/r03w_pytest_slow.log 2>&1; echo "slow parity rc=$?"; tail -3 gpurun_out/r03w_pytest_slow.log
ZKLC_P2_ADDMANY=multi timeout 1500 python -m pytest tests/test_gpu_plonky2.py -x -q -k "mainnet_signature" > gpurun_out/r03w_pytest_slow_addmany_multi.log 2>&1; echo "slow parity (multi) rc=$?"; tail -3 gpurun_out/r03w_pytest_slow_addmany_multi.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
