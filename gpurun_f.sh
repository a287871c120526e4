mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-bn254-extras --no-cpu-baseline --detail gpurun_out/r04o_bench_2ranks_one_gpu_detail.json > gpurun_out/r04o_bench_2ranks_one_gpu_line.json 2> gpurun_out/r04o_bench_2ranks.err; echo "2-rank rc=$?"; tail -3 gpurun_out/r04o_bench_2ranks.err | cut -c1-300
timeout 900 python bench.py --steps 1 --warmup 1 --c5-validators 1000 --no-bn254-extras --no-cpu-baseline --detail gpurun_out/r04o_bench_c5_1000_detail.json > gpurun_out/r04o_bench_c5_1000_line.json 2> gpurun_out/r04o_bench_c5.err; echo "c5 rc=$?"; tail -3 gpurun_out/r04o_bench_c5.err | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_goldilocks.py tests/test_gpu_bn254.py tests/test_gpu_recursion.py tests/test_gpu_witness.py -x -q > gpurun_out/r04o_pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r04o_pytest_subset.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cat gpurun_out/r04o_bench_2ranks_one_gpu_line.json | cut -c1-600; echo; python -c "
import json; l=json.loads(open('gpurun_out/r04o_bench_2ranks_one_gpu_line.json').read()); print(l['block_i']); print(l['stages']['msm'])
l=json.loads(open('gpurun_out/r04o_bench_c5_1000_line.json').read()); print(l['stages'].get('c5'))"
