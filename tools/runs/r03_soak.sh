#!/bin/bash
O=gpurun_out/r03soak; mkdir -p $O
export VRT_POOL_MIN_RAYS=0
timeout -k 10 400 python tests/soak/soak_scenes.py 100000 102500 > $O/a.log 2>&1; tail -2 $O/a.log
VRT_POOL_T_HIT=8 VRT_POOL_T_END=12 VRT_POOL_SWAP_MIN=1 VRT_POOL_REFILL_MIN=1 timeout -k 10 300 python tests/soak/soak_scenes.py 102500 104000 > $O/b.log 2>&1; tail -2 $O/b.log
VRT_POOL_T_HIT=100 VRT_POOL_T_END=90 VRT_POOL_SWAP_MIN=20 VRT_POOL_REFILL_MIN=40 VRT_POOL_KEEP=1 VRT_POOL_ITERS=9 VRT_MARCH_GRID=3 timeout -k 10 300 python tests/soak/soak_scenes.py 104000 105500 > $O/c.log 2>&1; tail -2 $O/c.log
grep -c FAILED $O/a.log $O/b.log $O/c.log
