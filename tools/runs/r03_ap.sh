#!/bin/bash
O=gpurun_out/r03ap; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
V=$PWD/python_raytracer_amd
run c3_t4 A=1; run c3_t1 VRT_SO=$V/_vrt_tail1.so; run c3_t8 VRT_SO=$V/_vrt_tail8.so; run c3_t4b A=1; run c3_t1b VRT_SO=$V/_vrt_tail1.so
CFG=x3 STEPS=20 run x3_t4 A=1; CFG=x3 STEPS=20 run x3_t1 VRT_SO=$V/_vrt_tail1.so; CFG=x3 STEPS=20 run x3_t8 VRT_SO=$V/_vrt_tail8.so
CFG=c2 STEPS=20 run c2pool_t4 VRT_POOL_MIN_RAYS=0; CFG=c2 STEPS=20 run c2pool_t1 VRT_POOL_MIN_RAYS=0 VRT_SO=$V/_vrt_tail1.so; CFG=c2 STEPS=20 run c2pool_t8 VRT_POOL_MIN_RAYS=0 VRT_SO=$V/_vrt_tail8.so; CFG=c2 STEPS=20 run c2lanes A=1
for v in "" _tail1 _tail8; do echo "share 1/8 $v"; env ${v:+VRT_SO=$V/_vrt$v.so} EXP_WORLDS=8 timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world; done
CFG=c5 STEPS=3 WARM=1 run c5_t4 A=1; CFG=c5 STEPS=3 WARM=1 run c5_t1 VRT_SO=$V/_vrt_tail1.so
