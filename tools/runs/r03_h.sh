#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
VRT_POOL=0 python tools/sweep_pool.py c5 -,-,-,-,-,- 2>&1 | tee $O/sweep_c5.txt
python tools/sweep_pool.py c5 48,40,4,8,40,5 48,32,4,8,40,5 48,24,4,8,40,5 44,40,4,8,40,5 52,40,4,8,40,5 48,40,4,8,44,5 48,40,4,8,48,4 48,40,4,8,40,3 48,40,6,8,40,5 48,40,4,16,40,5 48,40,4,4,40,5 2>&1 | tee -a $O/sweep_c5.txt
VRT_POOL=0 python tools/sweep_pool.py c3 -,-,-,-,-,- 2>&1 | tee $O/sweep_c3.txt
python tools/sweep_pool.py c3 40,60,8,8,40,3 40,64,8,8,40,3 36,60,8,8,40,3 40,60,8,16,48,3 40,60,6,8,40,3 40,60,8,8,44,4 2>&1 | tee -a $O/sweep_c3.txt
VRT_POOL=0 python tools/sweep_pool.py c5 -,-,-,-,-,- 2>&1 | tee -a $O/sweep_c5.txt
