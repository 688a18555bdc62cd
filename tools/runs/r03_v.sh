#!/bin/bash
O=gpurun_out/r03v; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
run c3_now A=1
for v in ilp memcl iter bias0 o2; do run c3_$v VRT_SO=$PWD/python_raytracer_amd/_vrt_$v.so; done
run c3_now2 A=1
CFG=c5 STEPS=3 WARM=1 run c5_now A=1
for v in ilp memcl iter; do CFG=c5 STEPS=3 WARM=1 run c5_$v VRT_SO=$PWD/python_raytracer_amd/_vrt_$v.so; done
VRT_DIAG=1 python3 tools/diag_march.py c3 2>&1 | grep -v amdgpu.ids > $O/diag_c3.txt; VRT_DIAG=1 python3 tools/diag_march.py c5 2>&1 | grep -v amdgpu.ids > $O/diag_c5.txt; grep "cycle shares\|timeline" $O/diag_c3.txt $O/diag_c5.txt
