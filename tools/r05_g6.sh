set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bn254.py -m gpu -x -q -k "fixed or tiny or uniform" > gpurun_out/r05f_pytest_msm_fixed.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r05f_pytest_msm_fixed.log
timeout 600 python tools/msm_quickbench.py 16 18 20 22 --fixed --dist=U,W > gpurun_out/r05f_msm_fixed_quickbench.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05f_msm_fixed_quickbench.txt | cut -c1-200
rm -rf gpurun_out/prof_tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o msm -- python tools/msm_quickbench.py 22 --fixed > gpurun_out/r05f_msm_trace.log 2>&1
find gpurun_out/prof_tmp -name '*kernel_stats.csv' -exec cp {} gpurun_out/r05f_msm_2p22_fixed_kernel_stats.csv \;
rm -rf gpurun_out/prof_tmp
cut -c1-70,180-260 gpurun_out/r05f_msm_2p22_fixed_kernel_stats.csv | head -24
