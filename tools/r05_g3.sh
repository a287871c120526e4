set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_goldilocks.py "tests/test_gpu_plonky2.py::test_every_quotient_evaluator_variant_gives_the_same_proof_bytes" "tests/test_gpu_plonky2.py::test_gpu_proof_bytes_equal_the_c_prover_at_reference_sizes" "tests/test_gpu_plonky2.py::test_all_gate_types_match_oracle_bit_for_bit" tests/test_gpu_recursion.py -m gpu -x -q > gpurun_out/r05c_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05c_pytest.log
rm -rf gpurun_out/prof_tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o prove -- python tools/prove_profile_ed25519.py 5 > gpurun_out/r05c_prove_ed25519.log 2>&1; echo "rocprof rc=$?"
find gpurun_out/prof_tmp -name '*kernel_stats.csv' -exec cp {} gpurun_out/r05c_prove_ed25519_kernel_stats.csv \;
rm -rf gpurun_out/prof_tmp
tail -2 gpurun_out/r05c_prove_ed25519.log | cut -c1-400
cut -c1-90 gpurun_out/r05c_prove_ed25519_kernel_stats.csv | head -5
