#!/bin/bash
O=gpurun_out/r03y; mkdir -p $O
export VRT_POOL_MIN_RAYS=0
VRT_POOL=0 python tools/sweep_pool.py c2 -,-,-,-,-,- 2>&1 | tee $O/sweep_c2.txt
python tools/sweep_pool.py c2 -,-,-,-,-,- 32,48,8,8,40,3 48,60,8,8,40,3 40,48,8,8,40,3 40,72,8,8,40,3 40,60,4,8,40,3 40,60,8,4,40,3 40,60,8,16,40,3 40,60,8,8,32,4 40,60,8,8,48,2 32,40,4,4,32,4 24,32,4,4,32,4 2>&1 | tee -a $O/sweep_c2.txt
VRT_POOL=0 python tools/sweep_pool.py c2 -,-,-,-,-,- 2>&1 | tee -a $O/sweep_c2.txt
for c in 64 128 256; do VRT_CHUNK=$c python tools/sweep_pool.py c2 -,-,-,-,-,- 2>&1 | sed "s/^/chunk $c /" | tee -a $O/sweep_c2.txt; done
