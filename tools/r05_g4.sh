set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/host_cpu_probe.py 17 10 > gpurun_out/r05d_host_cpu_probe.txt 2>&1; tail -4 gpurun_out/r05d_host_cpu_probe.txt | cut -c1-300
rm -rf /tmp/zklc_cold_cache
timeout 900 python tools/cold_start_profile.py /tmp/zklc_cold_cache > gpurun_out/r05d_cold_start_profile_cold.txt 2>&1; head -3 gpurun_out/r05d_cold_start_profile_cold.txt | cut -c1-300
timeout 900 python tools/cold_start_profile.py /tmp/zklc_cold_cache > gpurun_out/r05d_cold_start_profile_warm_cache.txt 2>&1; head -3 gpurun_out/r05d_cold_start_profile_warm_cache.txt | cut -c1-300
du -sh /tmp/zklc_cold_cache; ls -la /tmp/zklc_cold_cache | head -20
