#!/bin/bash
O=gpurun_out/r03x; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
PREV=$PWD/python_raytracer_amd/_vrt_prev.so
run c3_now A=1; run c3_prev VRT_SO=$PREV; run c3_now2 A=1; run c3_prev2 VRT_SO=$PREV
CFG=c5 STEPS=3 WARM=1 run c5_now A=1; CFG=c5 STEPS=3 WARM=1 run c5_prev VRT_SO=$PREV
CFG=c2 run c2_pool_now VRT_POOL_MIN_RAYS=0; CFG=c2 run c2_pool_prev VRT_POOL_MIN_RAYS=0 VRT_SO=$PREV; CFG=c2 run c2_lanes A=1
for w in 8 4; do
echo "== share 1/$w now / prev"; EXP_WORLDS=$w timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world; EXP_WORLDS=$w VRT_SO=$PREV timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world
done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or compact or knobs or random_scenes or retrace or third or axis or edge" --timeout 200 > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
