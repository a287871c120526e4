#!/bin/bash
# round 6, second GPU job: the pinned-staging waits and the circuit containers on the GPU
set -u
TAG=r06b; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_c_abi.py -x -q -s > gpurun_out/${TAG}_pytest_c_abi.log 2>&1; echo "c_abi rc=$?"; tail -5 gpurun_out/${TAG}_pytest_c_abi.log
timeout 300 python tools/host_cpu_probe.py 17 10 > gpurun_out/${TAG}_host_cpu_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/${TAG}_host_cpu_probe.txt | tail -4
timeout 120 python tools/repro_blocking_sync_hang.py > gpurun_out/${TAG}_repro_blocking_sync.txt 2>&1; echo "repro rc=$?" | tee -a gpurun_out/${TAG}_repro_blocking_sync.txt; tail -6 gpurun_out/${TAG}_repro_blocking_sync.txt
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/${TAG}_bench_detail.json; tail -c 1500 gpurun_out/${TAG}_bench_line.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/${TAG}_pytest_gpu.log
