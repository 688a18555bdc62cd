#!/bin/bash
O=gpurun_out/r03ak; mkdir -p $O
timeout -k 10 500 python -m pytest tests -q -m gpu -x --timeout 200 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
export VRT_POOL_MIN_RAYS=0
timeout -k 10 300 python tests/soak/soak_scenes.py 130000 131500 > $O/a.log 2>&1; tail -1 $O/a.log
VRT_POOL_T_HIT=8 VRT_POOL_T_END=12 VRT_POOL_SWAP_MIN=1 VRT_POOL_REFILL_MIN=1 timeout -k 10 300 python tests/soak/soak_scenes.py 131500 132500 > $O/b.log 2>&1; tail -1 $O/b.log
VRT_POOL_T_HIT=100 VRT_POOL_T_END=90 VRT_POOL_SWAP_MIN=20 VRT_POOL_REFILL_MIN=40 VRT_POOL_KEEP=1 VRT_POOL_ITERS=9 VRT_MARCH_GRID=3 timeout -k 10 300 python tests/soak/soak_scenes.py 132500 133500 > $O/c.log 2>&1; tail -1 $O/c.log
