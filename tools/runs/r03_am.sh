#!/bin/bash
O=gpurun_out/r03am; mkdir -p $O
timeout -k 10 300 python tests/soak/soak_select.py > $O/select.log 2>&1; tail -1 $O/select.log
timeout -k 10 400 python tests/soak/soak_scenes.py 140000 143000 > $O/lanes.log 2>&1; tail -1 $O/lanes.log
VRT_POOL_MIN_RAYS=0 timeout -k 10 400 python tests/soak/soak_scenes.py 143000 146000 > $O/pool.log 2>&1; tail -1 $O/pool.log
