#!/bin/bash
# round 3, GPU call y: NTT tile size A/B with the shift-twiddle passes (2^13-element tiles, two passes at 2^21, against 2^12, three)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/ntt_quickbench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03y_ntt_tile_default.txt
ZKLC_NTT_TILE=12 timeout 200 python tools/ntt_quickbench.py 2>&1 | grep -v amdgpu | sed 's/^/tile12: /' | tee gpurun_out/r03y_ntt_tile12.txt
