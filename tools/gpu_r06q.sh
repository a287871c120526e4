#!/bin/bash
# round 6, GPU job q: counters of the tree with the zero-aware LDE: block-level accounting (trace + two PMC passes), LDE PMC at both shapes
set -u
TAG=r06q; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_block_accounting.sh ${TAG} 4 2 > gpurun_out/${TAG}_accounting_script.log 2>&1; head -5 gpurun_out/${TAG}_accounting_script.log
bash tools/pmc_lde.sh ${TAG}c3 17 > gpurun_out/${TAG}_pmc_lde_c3_script.log 2>&1; tail -1 gpurun_out/${TAG}_pmc_lde_c3_script.log | cut -c1-300
bash tools/pmc_lde.sh ${TAG}p18 18 > gpurun_out/${TAG}_pmc_lde_p18_script.log 2>&1; tail -1 gpurun_out/${TAG}_pmc_lde_p18_script.log | cut -c1-300
rm -f gpurun_out/${TAG}*_pmc_lde_*.csv
