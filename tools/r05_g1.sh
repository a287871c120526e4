set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rccl_world1.py tests/test_gpu_goldilocks.py "tests/test_gpu_plonky2.py::test_every_quotient_evaluator_variant_gives_the_same_proof_bytes" "tests/test_gpu_plonky2.py::test_gpu_proof_bytes_equal_the_c_prover_at_reference_sizes" -m gpu -x -q > gpurun_out/r05a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05a_pytest.log
timeout 120 tools/ubench/valu_ubench > gpurun_out/r05a_valu_ubench.txt 2>&1; echo "ubench rc=$?"; head -8 gpurun_out/r05a_valu_ubench.txt
timeout 900 bash tools/pmc_lde.sh r05a_2p18 18; echo "pmc rc=$?"
rm -rf gpurun_out/prof_tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o lde -- python tools/lde_only.py 10 18 > gpurun_out/r05a_lde18_trace.log 2>&1
find gpurun_out/prof_tmp -name '*kernel_stats.csv' -exec cp {} gpurun_out/r05a_lde_2p18_kernel_stats.csv \;
rm -rf gpurun_out/prof_tmp
cat gpurun_out/r05a_lde_2p18_kernel_stats.csv | head -8
