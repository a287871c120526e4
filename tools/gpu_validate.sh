#!/bin/bash
# The closing validation of a round on the GPU box (via gpurun): the driver's bench command from a COLD circuit cache FIRST (it is the
# number that matters; round 5 ran it last and lost it), then smoke(), the block-level accounting (kernel trace + PMC passes) and the
# PMC of the dominant kernel on the final tree, the rocprofv3 kernel summary of five Ed25519-circuit proofs, and the full GPU suite.
# Everything lands under gpurun_out/ and is copied into profiles/ by hand afterwards.        bash tools/gpu_validate.sh r06h
set -u
TAG=${1:-r06h}
mkdir -p gpurun_out; export TMPDIR=/tmp
uptime > gpurun_out/${TAG}_host.txt; nproc >> gpurun_out/${TAG}_host.txt
rm -rf .circuit_cache
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd_line.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/${TAG}_bench_driver_cmd_detail.json
tail -c 1500 gpurun_out/${TAG}_bench_driver_cmd_line.json; echo
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${TAG}_smoke.log
bash tools/gpu_block_accounting.sh ${TAG} 4 2 > gpurun_out/${TAG}_accounting_script.log 2>&1; head -5 gpurun_out/${TAG}_accounting_script.log
bash tools/pmc_merkle.sh ${TAG} > gpurun_out/${TAG}_pmc_merkle_script.log 2>&1; tail -1 gpurun_out/${TAG}_pmc_merkle_script.log | cut -c1-400
rm -f gpurun_out/${TAG}_pmc_merkle_*.csv
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o prove -- python tools/prove_profile_ed25519.py 5 > gpurun_out/${TAG}_prove_ed25519.log 2>&1; echo "rocprof prove rc=$?"
find gpurun_out/prof_tmp -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_prove_ed25519_kernel_stats.csv \;
rm -rf gpurun_out/prof_tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_pytest_gpu.log
