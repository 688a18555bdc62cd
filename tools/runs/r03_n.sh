#!/bin/bash
for w in 8 4 1; do
echo "== share 1/$w"; EXP_WORLDS=$w VRT_POOL_MIN_RAYS=0 timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world; EXP_WORLDS=$w VRT_POOL=0 timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world
done
for c in 64 128 256; do echo "chunk $c"; EXP_WORLDS=8 VRT_CHUNK=$c VRT_POOL_MIN_RAYS=0 timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world; done
