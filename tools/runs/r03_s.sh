#!/bin/bash
O=gpurun_out/r03s; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['roofline']['kernel'])"
}
run c3_now A=1; run c3_v1 VRT_SO=$PWD/python_raytracer_amd/_vrt_v1.so; run c3_nopp VRT_SO=$PWD/python_raytracer_amd/_vrt_nopp.so; run c3_nostall VRT_SO=$PWD/python_raytracer_amd/_vrt_nostall.so; run c3_noboth VRT_SO=$PWD/python_raytracer_amd/_vrt_noboth.so; run c3_v1b VRT_SO=$PWD/python_raytracer_amd/_vrt_v1.so
