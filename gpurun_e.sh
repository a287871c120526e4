mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/pmc_lde.sh r04n > gpurun_out/r04n_pmc_lde_summary.txt 2>&1
bash tools/pmc_merkle.sh r04n > gpurun_out/r04n_pmc_merkle_summary.txt 2>&1
[ -s gpurun_out/r04n_lde_pmc.json ] && cp gpurun_out/r04n_lde_pmc.json profiles/lde_pmc_latest.json
[ -s gpurun_out/r04n_poseidon_pmc.json ] && cp gpurun_out/r04n_poseidon_pmc.json profiles/poseidon_pmc_latest.json
rm -f gpurun_out/r04n_pmc_lde_*.csv gpurun_out/r04n_pmc_merkle_*.csv
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r04n_bench_driver_cmd_detail.json > gpurun_out/r04n_bench_driver_cmd_line.json 2> gpurun_out/r04n_bench.err; echo "bench rc=$?"; wc -c gpurun_out/r04n_bench_driver_cmd_line.json
rm -rf gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bn254-extras --detail gpurun_out/r04n_bench_prof_detail.json > gpurun_out/r04n_bench_prof_line.json 2>/dev/null; echo "rocprof rc=$?"
find gpurun_out/prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/r04n_bench_kernel_stats.csv \;
rm -rf gpurun_out/prof
tail -1 gpurun_out/r04n_pmc_lde_summary.txt | cut -c1-300; tail -1 gpurun_out/r04n_pmc_merkle_summary.txt | cut -c1-300
cat gpurun_out/r04n_bench_driver_cmd_line.json | cut -c1-1500
