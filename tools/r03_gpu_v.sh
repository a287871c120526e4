#!/bin/bash
# round 3, GPU call v: all U32AddMany variants in one evaluator: parity (gate types bit for bit, C-prover bytes, the real Ed25519
# circuit proof accepted by the verifier) and the per-kernel split of the Ed25519-circuit proof
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_plonky2.py -x -q -k "bit_for_bit or c_prover or mainnet_signature" > gpurun_out/r03v_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03v_pytest.log
rm -rf gpurun_out/prof_tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -o p -- python tools/prove_profile_ed25519.py 3 > gpurun_out/r03v_prove_ed25519.log 2>&1; echo "rc=$?"; grep "wires_commit" gpurun_out/r03v_prove_ed25519.log | cut -c1-400
f=$(find gpurun_out/prof_tmp -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r03v_prove_ed25519_kernel_stats.csv
rm -rf gpurun_out/prof_tmp
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r03v_prove_ed25519_kernel_stats.csv")))
for r in rows[:14]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), "%9.3f ms" % (float(r["TotalDurationNs"]) / 1e6), "%9.1f us" % (float(r["AverageNs"]) / 1e3))
PY
