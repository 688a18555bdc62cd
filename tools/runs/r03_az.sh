#!/bin/bash
O=gpurun_out/r03az; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -x --timeout 300 > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
VRT_POOL_MIN_RAYS=0 timeout -k 10 200 python tests/soak/soak_scenes.py 300000 303000 > $O/pool.log 2>&1; tail -1 $O/pool.log
timeout -k 10 200 python tests/soak/soak_scenes.py 303000 306000 > $O/lanes.log 2>&1; tail -1 $O/lanes.log
bash tools/profile_all.sh r03_v5 > gpurun_out/profile_all.log 2>&1; tail -1 gpurun_out/profile_all.log
