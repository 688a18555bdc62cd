#!/bin/bash
# SQ counters of the pool march against the one-ray-per-lane march (refactored), config 3
set -o pipefail
export VRT_POOL=1; bash tools/pmc_run.sh r03b_c3_pool1 "--config c3" || exit 1
export VRT_POOL=0; bash tools/pmc_run.sh r03b_c3_pool0 "--config c3" || exit 1
grep -A26 "march_pool_kernel<8, 1>\|march_kernel<8, 1, false, false, 0>" gpurun_out/pmc_r03b_c3_pool1_summary.txt gpurun_out/pmc_r03b_c3_pool0_summary.txt
