#!/bin/bash
set -u
TAG=${1:-r02a}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_pytest_gpu.log
tail -25 gpurun_out/${TAG}_pytest_gpu.log
tools/pmc_sq.sh ${TAG} python tools/gl_quickbench.py
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/${TAG}_bench.err
head -c 1500 gpurun_out/${TAG}_bench.json
