#!/bin/bash
# Block-level accounting of the headline on the GPU box (via gpurun): the overlapped bench under rocprofv3 --kernel-trace, then two
# PMC passes (SQ_INSTS_VALU; GRBM_GUI_ACTIVE + SQ_BUSY_CYCLES) of the same command, reduced by tools/block_accounting.py.
#        bash tools/gpu_block_accounting.sh r06a [steps] [warmup]
set -u
TAG=${1:-r06a}; STEPS=${2:-4}; WARM=${3:-2}
mkdir -p gpurun_out; export TMPDIR=/tmp
BENCH="python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --c5-validators 0 --no-bn254-extras"
rm -rf gpurun_out/acc_tmp
ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_trace_detail.json timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/acc_tmp -o bench -- $BENCH > gpurun_out/${TAG}_trace_bench.log 2>&1; echo "trace rc=$?"
find gpurun_out/acc_tmp -name '*kernel_stats.csv' -exec cp {} gpurun_out/${TAG}_bench_kernel_stats.csv \;
TR=$(find gpurun_out/acc_tmp -name '*kernel_trace.csv' | head -1)
for grp in "SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  n=$(echo $grp | cut -d' ' -f1)
  ZKLC_BENCH_DETAIL=gpurun_out/${TAG}_pmc_${n}_detail.json timeout 1500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/acc_pmc_$n -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --c5-validators 0 --no-bn254-extras > gpurun_out/${TAG}_pmc_${n}.log 2>&1; echo "pmc $n rc=$?"
done
SPEC=""
for n in SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
  C=$(find gpurun_out/acc_pmc_$n -name '*counter_collection.csv' | head -1); K=$(find gpurun_out/acc_pmc_$n -name '*kernel_trace.csv' | head -1)
  [ -n "$C" ] && SPEC="$SPEC $C,gpurun_out/${TAG}_pmc_${n}_detail.json${K:+,$K}"
done
python tools/block_accounting.py --trace "$TR" --detail gpurun_out/${TAG}_trace_detail.json --pmc $SPEC \
   --out gpurun_out/${TAG}_block_accounting.json --compact gpurun_out/${TAG}_trace_compact.npz > gpurun_out/${TAG}_block_accounting.log 2>&1; echo "accounting rc=$?"
tail -c 3000 gpurun_out/${TAG}_block_accounting.log
head -3 "$C" > gpurun_out/${TAG}_pmc_csv_head.txt
rm -rf gpurun_out/acc_tmp gpurun_out/acc_pmc_SQ_INSTS_VALU gpurun_out/acc_pmc_GRBM_GUI_ACTIVE
