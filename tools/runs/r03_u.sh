#!/bin/bash
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 500 python -m pytest tests -q -m gpu -x --timeout 200 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
export VRT_POOL_MIN_RAYS=0
timeout -k 10 300 python tests/soak/soak_scenes.py 110000 111500 > $O/a.log 2>&1; tail -1 $O/a.log
VRT_POOL_T_HIT=8 VRT_POOL_T_END=12 VRT_POOL_SWAP_MIN=1 VRT_POOL_REFILL_MIN=1 timeout -k 10 300 python tests/soak/soak_scenes.py 111500 112500 > $O/b.log 2>&1; tail -1 $O/b.log
unset VRT_POOL_MIN_RAYS
bash tools/profile_all.sh r03_v2 > gpurun_out/profile_all.log 2>&1; tail -1 gpurun_out/profile_all.log
