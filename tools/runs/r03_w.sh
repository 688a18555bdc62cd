#!/bin/bash
O=gpurun_out/r03w; mkdir -p $O
for cfg in c3 c5; do VRT_DIAG=1 python3 tools/diag_march.py $cfg 2>&1 | grep -v amdgpu.ids > $O/diag_$cfg.txt; cat $O/diag_$cfg.txt; done
VRT_POOL=0 VRT_DIAG=1 python3 tools/diag_march.py c3 2>&1 | grep -v amdgpu.ids > $O/diag_c3_lanes.txt; cat $O/diag_c3_lanes.txt
