#!/bin/bash
O=gpurun_out/r03aw; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
V=$PWD/python_raytracer_amd
run c3_nt A=1; run c3_nont VRT_SO=$V/_vrt_nont.so; run c3_nt2 A=1; run c3_nont2 VRT_SO=$V/_vrt_nont.so
CFG=c5 STEPS=3 WARM=1 run c5_nt A=1; CFG=c5 STEPS=3 WARM=1 run c5_nont VRT_SO=$V/_vrt_nont.so
CFG=c2 STEPS=20 run c2_nt A=1; CFG=c2 STEPS=20 run c2_nont VRT_SO=$V/_vrt_nont.so
CFG=x1 run x1_nt A=1; CFG=x1 run x1_nont VRT_SO=$V/_vrt_nont.so
