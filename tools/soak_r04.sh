#!/bin/bash
# usage (via gpurun, from the repo root): bash tools/soak_r04.sh N   -- N random scenes per regime against the oracle
N=${1:-2000}; O=gpurun_out/soak_r04; mkdir -p $O
export VRT_POOL_MIN_RAYS=0
SOAK_MAXRES=2 VRT_WADDR=1 VRT_POOL=1 timeout -k 10 900 python tests/soak/soak_scenes.py 100000 $((100000+N)) > $O/ahead_pool.log 2>&1
SOAK_MAXRES=2 VRT_WADDR=1 VRT_POOL=0 timeout -k 10 900 python tests/soak/soak_scenes.py 200000 $((200000+N)) > $O/ahead_lanes.log 2>&1
VRT_POOL=1 timeout -k 10 900 python tests/soak/soak_scenes.py 300000 $((300000+N)) > $O/pool.log 2>&1
VRT_POOL=0 timeout -k 10 900 python tests/soak/soak_scenes.py 400000 $((400000+N)) > $O/lanes.log 2>&1
tail -n 2 $O/*.log
# the kernel instances that compare a re-snap's traversed key behind the voxel reads (scenes beyond the caches get them;
# VRT_DEFER_VISIT=2 VRT_TRAV_LDS=0 puts these small scenes on them)
if [ -n "$SOAK_DEFER" ]; then
  VRT_TRAV_LDS=0 VRT_DEFER_VISIT=2 SOAK_MAXRES=2 VRT_POOL=1 timeout -k 10 500 python tests/soak/soak_scenes.py 500000 $((500000+SOAK_DEFER)) > $O/defer_pool.log 2>&1
  VRT_TRAV_LDS=0 VRT_DEFER_VISIT=2 SOAK_MAXRES=2 VRT_POOL=0 timeout -k 10 500 python tests/soak/soak_scenes.py 600000 $((600000+SOAK_DEFER)) > $O/defer_lanes.log 2>&1
  tail -n 2 $O/defer_*.log
fi
# frames without cached tables: the march's lanes make their own ray records (take_ray, PERPIX 3)
if [ -n "$SOAK_UNCACHED_N" ]; then
  SOAK_UNCACHED=1 SOAK_MAXRES=2 VRT_POOL=1 timeout -k 10 500 python tests/soak/soak_scenes.py 700000 $((700000+SOAK_UNCACHED_N)) > $O/uncached_pool.log 2>&1
  SOAK_UNCACHED=1 SOAK_MAXRES=2 VRT_POOL=0 timeout -k 10 500 python tests/soak/soak_scenes.py 800000 $((800000+SOAK_UNCACHED_N)) > $O/uncached_lanes.log 2>&1
  tail -n 2 $O/uncached_*.log
fi
# boxes with the settled bitmap over the 32^3 cells around the camera only (what boxes too large for a bitmap get), with and
# without the key comparison behind the voxel reads
if [ -n "$SOAK_WINDOW_N" ]; then
  VRT_TRAV_WINDOW=2 VRT_POOL=1 timeout -k 10 400 python tests/soak/soak_scenes.py 900000 $((900000+SOAK_WINDOW_N)) > $O/window_pool.log 2>&1
  VRT_TRAV_WINDOW=2 VRT_POOL=0 timeout -k 10 400 python tests/soak/soak_scenes.py 910000 $((910000+SOAK_WINDOW_N)) > $O/window_lanes.log 2>&1
  VRT_TRAV_WINDOW=2 VRT_DEFER_VISIT=2 SOAK_MAXRES=2 VRT_POOL=1 timeout -k 10 400 python tests/soak/soak_scenes.py 920000 $((920000+SOAK_WINDOW_N)) > $O/window_defer_pool.log 2>&1
  VRT_TRAV_WINDOW=2 VRT_DEFER_VISIT=2 SOAK_MAXRES=2 VRT_POOL=0 timeout -k 10 400 python tests/soak/soak_scenes.py 930000 $((930000+SOAK_WINDOW_N)) > $O/window_defer_lanes.log 2>&1
  tail -n 2 $O/window_*.log
fi
