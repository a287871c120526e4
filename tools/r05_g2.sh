set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r05b_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r05b_pytest_gpu.log
timeout 1200 python bench.py --steps 6 --warmup 2 > gpurun_out/r05b_bench_line.json 2> gpurun_out/r05b_bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/r05b_bench_detail.json
tail -c 1500 gpurun_out/r05b_bench_line.json
