#!/bin/bash
O=gpurun_out/r03at; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
V=$PWD/python_raytracer_amd
CFG=c5 STEPS=3 WARM=1 run c5_now A=1; CFG=c5 STEPS=3 WARM=1 run c5_ident VRT_SO=$V/_vrt_ident.so; CFG=c5 STEPS=3 WARM=1 run c5_now2 A=1; CFG=c5 STEPS=3 WARM=1 run c5_ident2 VRT_SO=$V/_vrt_ident.so
CFG=c5 STEPS=3 WARM=1 run c5_now_lanes VRT_POOL=0; CFG=c5 STEPS=3 WARM=1 run c5_ident_lanes VRT_POOL=0 VRT_SO=$V/_vrt_ident.so
