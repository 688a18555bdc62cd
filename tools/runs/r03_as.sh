#!/bin/bash
O=gpurun_out/r03as; mkdir -p $O
timeout -k 10 250 python tests/soak/soak_select.py 50000 53000 > $O/select.log 2>&1; tail -1 $O/select.log
timeout -k 10 300 python tests/soak/soak_scenes.py 200000 208000 > $O/lanes.log 2>&1; tail -1 $O/lanes.log
export VRT_POOL_MIN_RAYS=0
timeout -k 10 300 python tests/soak/soak_scenes.py 208000 216000 > $O/pool_a.log 2>&1; tail -1 $O/pool_a.log
VRT_POOL_T_HIT=8 VRT_POOL_T_END=12 VRT_POOL_SWAP_MIN=1 VRT_POOL_REFILL_MIN=1 timeout -k 10 300 python tests/soak/soak_scenes.py 216000 222000 > $O/pool_b.log 2>&1; tail -1 $O/pool_b.log
VRT_POOL_T_HIT=100 VRT_POOL_T_END=90 VRT_POOL_SWAP_MIN=20 VRT_POOL_REFILL_MIN=40 VRT_POOL_KEEP=1 VRT_POOL_ITERS=9 VRT_MARCH_GRID=3 timeout -k 10 300 python tests/soak/soak_scenes.py 222000 228000 > $O/pool_c.log 2>&1; tail -1 $O/pool_c.log
