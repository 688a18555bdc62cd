#!/bin/bash
O=gpurun_out/r03t; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['roofline']['kernel'])"
}
run c3_now A=1; run c3_v1 VRT_SO=$PWD/python_raytracer_amd/_vrt_v1.so; run c3_now2 A=1; run c3_v1b VRT_SO=$PWD/python_raytracer_amd/_vrt_v1.so
CFG=c5 STEPS=3 WARM=1 run c5_now A=1; CFG=c5 STEPS=3 WARM=1 run c5_v1 VRT_SO=$PWD/python_raytracer_amd/_vrt_v1.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or compact or knobs or random_scenes or retrace or third or axis or edge or config5" --timeout 200 > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
