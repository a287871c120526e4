set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_goldilocks.py tests/test_gpu_bn254.py "tests/test_gpu_plonky2.py::test_every_quotient_evaluator_variant_gives_the_same_proof_bytes" "tests/test_gpu_plonky2.py::test_gpu_proof_bytes_equal_the_c_prover_at_reference_sizes" "tests/test_gpu_plonky2.py::test_all_gate_types_match_oracle_bit_for_bit" "tests/test_gpu_plonky2.py::test_proof_matches_oracle_bit_for_bit" tests/test_gpu_recursion.py tests/test_gpu_groth16.py -m gpu -x -q > gpurun_out/r05g_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05g_pytest.log
timeout 600 python tools/msm_quickbench.py 20 22 --fixed --dist=U,W > gpurun_out/r05g_msm_fixed_quickbench.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05g_msm_fixed_quickbench.txt | grep -v oracle | cut -c1-200
timeout 300 python tools/prove_profile.py ed 18 6 > gpurun_out/r05g_prove_ed_synth18.txt 2>&1; tail -1 gpurun_out/r05g_prove_ed_synth18.txt | cut -c1-300
