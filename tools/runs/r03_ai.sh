#!/bin/bash
O=gpurun_out/r03ai; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
V=$PWD/python_raytracer_amd
run c3_now A=1; run c3_split4 VRT_SO=$V/_vrt_split4.so; run c3_split2 VRT_SO=$V/_vrt_split2.so; run c3_now2 A=1; run c3_split4b VRT_SO=$V/_vrt_split4.so
CFG=c5 STEPS=3 WARM=1 run c5_now A=1; CFG=c5 STEPS=3 WARM=1 run c5_split4 VRT_SO=$V/_vrt_split4.so; CFG=c5 STEPS=3 WARM=1 run c5_split2 VRT_SO=$V/_vrt_split2.so
CFG=c2 run c2_now A=1; CFG=c2 run c2_split4 VRT_SO=$V/_vrt_split4.so
