#!/bin/bash
O=gpurun_out/r03m; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['roofline']['kernel'])"
}
run c3_default A=1
run c3_s40 VRT_SO=$PWD/python_raytracer_amd/_vrt_pool40.so
run c3_s56 VRT_SO=$PWD/python_raytracer_amd/_vrt_pool56.so VRT_POOL_VERBOSE=1; grep "LDS per" $O/c3_s56.err | tail -2
run c3_fm1 VRT_SO=$PWD/python_raytracer_amd/_vrt_fm1.so
CFG=c2 run c2_default A=1; CFG=c2 run c2_pool VRT_POOL_MIN_RAYS=0
CFG=c2 run c2_pool_t VRT_POOL_MIN_RAYS=0 VRT_POOL_T_HIT=32 VRT_POOL_T_END=40
for w in 8 4; do
echo "== share 1/$w"; EXP_WORLDS=$w VRT_POOL_MIN_RAYS=0 timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world; EXP_WORLDS=$w VRT_POOL=0 timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world
done
CFG=c5 STEPS=3 WARM=1 run c5_default A=1
CFG=c5 STEPS=3 WARM=1 run c5_fm1 VRT_SO=$PWD/python_raytracer_amd/_vrt_fm1.so
