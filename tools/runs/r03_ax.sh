#!/bin/bash
O=gpurun_out/r03ax; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
V=$PWD/python_raytracer_amd
for i in 1 2 3; do CFG=c2 STEPS=30 run c2_nt$i A=1; CFG=c2 STEPS=30 run c2_nont$i VRT_SO=$V/_vrt_nont.so; done
run c3_nt A=1; run c3_nont VRT_SO=$V/_vrt_nont.so; run c3_nt2 A=1; run c3_nont2 VRT_SO=$V/_vrt_nont.so
run c3_lanes_nt VRT_POOL=0; run c3_lanes_nont VRT_POOL=0 VRT_SO=$V/_vrt_nont.so
CFG=x3 STEPS=20 run x3_lanes_nt VRT_POOL=0; CFG=x3 STEPS=20 run x3_lanes_nont VRT_POOL=0 VRT_SO=$V/_vrt_nont.so
for w in 16 32; do echo "share 1/$w nt / nont"; EXP_WORLDS=$w timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world; EXP_WORLDS=$w VRT_SO=$V/_vrt_nont.so timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world; done
