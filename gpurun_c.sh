mkdir -p gpurun_out; export TMPDIR=/tmp
ZKLC_P2_ADDMANY=pergate timeout 400 python tools/addmany_ab.py 5 > gpurun_out/r04c_addmany_ab.txt 2>&1
timeout 300 python tools/addmany_ab.py 5 >> gpurun_out/r04c_addmany_ab.txt 2>&1
rm -rf gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ed -- python tools/addmany_ab.py 5 > /dev/null 2>&1
find gpurun_out/prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/r04c_prove_ed25519_kernel_stats.csv \;
rm -rf gpurun_out/prof
grep -v amdgpu.ids gpurun_out/r04c_addmany_ab.txt
head -12 gpurun_out/r04c_prove_ed25519_kernel_stats.csv | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_plonky2.py -x -q -k "oracle or ed25519 or parity or byte" > gpurun_out/r04c_pytest_plonky2.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04c_pytest_plonky2.log
bash tools/pmc_kernels.sh r04c_addmany p2_quotient_addmany 5 -- python tools/addmany_ab.py 5 > gpurun_out/r04c_pmc_summary.txt 2>&1; tail -1 gpurun_out/r04c_pmc_summary.txt
