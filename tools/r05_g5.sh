set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/host_cpu_probe.py 17 20 > gpurun_out/r05e_host_cpu_probe_blocking.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05e_host_cpu_probe_blocking.txt | head -2 | cut -c1-300
ZKLC_SPIN_WAIT=1 timeout 300 python tools/host_cpu_probe.py 17 20 > gpurun_out/r05e_host_cpu_probe_spin.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05e_host_cpu_probe_spin.txt | head -2 | cut -c1-300
rm -rf /tmp/zklc_cold_cache
timeout 900 python tools/cold_start_profile.py /tmp/zklc_cold_cache > gpurun_out/r05e_cold_start_with_prewarm.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05e_cold_start_with_prewarm.txt | head -3 | cut -c1-400
export ZKLC_CIRCUIT_CACHE=/tmp/zklc_cold_cache
timeout 1200 python -m pytest tests/test_gpu_stream_pipeline.py tests/test_gpu_sha256.py -m gpu -x -q > gpurun_out/r05e_pytest_pipeline.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05e_pytest_pipeline.log
