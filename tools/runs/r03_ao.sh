#!/bin/bash
O=gpurun_out/r03ao; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-20} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
for g in 1024 768 512 384 256; do CFG=c2 run c2_grid$g VRT_MARCH_GRID=$g; done
CFG=c2 run c2_grid1024b VRT_MARCH_GRID=1024
for g in 1024 768 512; do CFG=x3 run x3_grid$g VRT_MARCH_GRID=$g; done
for g in 1024 768 512; do echo "share 1/8 grid $g"; VRT_MARCH_GRID=$g EXP_WORLDS=8 timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world; done
