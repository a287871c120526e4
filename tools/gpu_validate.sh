#!/bin/bash
# The closing validation of a round on the GPU box (via gpurun): full GPU suite, smoke(), the driver's bench command from a COLD circuit
# cache, and the rocprofv3 kernel summaries of the bench command and of five Ed25519-circuit proofs.  Everything lands under gpurun_out/
# and is copied into profiles/ by hand afterwards.        bash tools/gpu_validate.sh r05h
set -u
TAG=${1:-r05h}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${TAG}_smoke.log
rm -rf .circuit_cache
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd_line.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/${TAG}_bench_driver_cmd_detail.json
tail -c 1200 gpurun_out/${TAG}_bench_driver_cmd_line.json
rm -rf gpurun_out/prof_tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --c5-validators 0 > gpurun_out/${TAG}_prof_bench.log 2>&1; echo "rocprof bench rc=$?"
find gpurun_out/prof_tmp -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_bench_kernel_stats.csv \;
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o prove -- python tools/prove_profile_ed25519.py 5 > gpurun_out/${TAG}_prove_ed25519.log 2>&1; echo "rocprof prove rc=$?"
find gpurun_out/prof_tmp -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv \;
rm -rf gpurun_out/prof_tmp
