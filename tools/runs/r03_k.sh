#!/bin/bash
O=gpurun_out/r03k; mkdir -p $O
for cfg in c3 c5; do
VRT_DIAG=1 timeout -k 10 400 python tools/diag_march.py $cfg > $O/diag_${cfg}_pool1.txt 2>&1; grep -v amdgpu.ids $O/diag_${cfg}_pool1.txt
done
export VRT_POOL=1; bash tools/pmc_run.sh r03k_c3_pool1 "--config c3" || exit 1
grep -A26 "march_pool_kernel<8, 1>" gpurun_out/pmc_r03k_c3_pool1_summary.txt
