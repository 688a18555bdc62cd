#!/bin/bash
O=gpurun_out/r03aa; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
V=$PWD/python_raytracer_amd
run c3_now A=1; run c3_spec6 VRT_SO=$V/_vrt_spec6.so; run c3_spec10 VRT_SO=$V/_vrt_spec10.so; run c3_spec12 VRT_SO=$V/_vrt_spec12.so; run c3_now2 A=1
CFG=c5 STEPS=3 WARM=1 run c5_now A=1; CFG=c5 STEPS=3 WARM=1 run c5_spec6 VRT_SO=$V/_vrt_spec6.so; CFG=c5 STEPS=3 WARM=1 run c5_spec10 VRT_SO=$V/_vrt_spec10.so; CFG=c5 STEPS=3 WARM=1 run c5_spec12 VRT_SO=$V/_vrt_spec12.so
VRT_DIAG=1 timeout -k 10 300 python tools/diag_march.py c3 > $O/diag_c3.txt 2>&1; tail -12 $O/diag_c3.txt
VRT_DIAG=1 timeout -k 10 400 python tools/diag_march.py c5 > $O/diag_c5.txt 2>&1; tail -12 $O/diag_c5.txt
