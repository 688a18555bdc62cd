#!/bin/bash
O=gpurun_out/r03ba; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
V=$PWD/python_raytracer_amd
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or compact or knobs or random_scenes or retrace or third or axis or edge" --timeout 200 > $O/pytest_subset.log 2>&1; tail -2 $O/pytest_subset.log
run c3_now A=1; run c3_prev VRT_SO=$V/_vrt_prev.so; run c3_now2 A=1; run c3_prev2 VRT_SO=$V/_vrt_prev.so; run c3_now3 A=1; run c3_prev3 VRT_SO=$V/_vrt_prev.so
CFG=c5 STEPS=3 WARM=1 run c5_now A=1; CFG=c5 STEPS=3 WARM=1 run c5_prev VRT_SO=$V/_vrt_prev.so
CFG=c2 STEPS=30 run c2_now A=1; CFG=c2 STEPS=30 run c2_prev VRT_SO=$V/_vrt_prev.so; CFG=c2 STEPS=30 run c2_now2 A=1; CFG=c2 STEPS=30 run c2_prev2 VRT_SO=$V/_vrt_prev.so
CFG=x1 run x1_now A=1; CFG=x1 run x1_prev VRT_SO=$V/_vrt_prev.so
