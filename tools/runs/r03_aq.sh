#!/bin/bash
O=gpurun_out/r03aq; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
for i in 1 2 3 4; do CFG=c2 STEPS=30 run c2_pool$i VRT_POOL_MIN_RAYS=0; CFG=c2 STEPS=30 run c2_lanes$i A=1; done
for i in 1 2; do CFG=c2 STEPS=30 run c2_pool_chunk64_$i VRT_POOL_MIN_RAYS=0 VRT_CHUNK=64;  CFG=c2 STEPS=30 run c2_pool_chunk256_$i VRT_POOL_MIN_RAYS=0 VRT_CHUNK=256; done
