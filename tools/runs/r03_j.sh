#!/bin/bash
set -o pipefail
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
run() { name=$1; shift; args=$1; shift
  env "$@" timeout -k 10 600 python bench.py $args > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['roofline']['kernel'], d['config']['image_sha256'][:12], d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('sample'))"
}
run c3_fixture "--config c3 --steps 10 --warmup 3 --no-cpu --no-context" A=1
run c3_world "--config c3 --steps 10 --warmup 3 --no-cpu --no-context --world-flow" A=1
run c3_world_pool0 "--config c3 --steps 10 --warmup 3 --no-cpu --no-context --world-flow" VRT_POOL=0
run c3_res2_spec4 "--config c3 --steps 10 --warmup 3 --no-cpu --no-context" VRT_RESMODE=2 VRT_SPEC_DEEP=0
run c3_res2_spec8 "--config c3 --steps 10 --warmup 3 --no-cpu --no-context" VRT_RESMODE=2 VRT_SPEC_DEEP=1
run c3_res2_spec4_pool0 "--config c3 --steps 10 --warmup 3 --no-cpu --no-context" VRT_RESMODE=2 VRT_SPEC_DEEP=0 VRT_POOL=0
run c3_res2_spec8_pool0 "--config c3 --steps 10 --warmup 3 --no-cpu --no-context" VRT_RESMODE=2 VRT_SPEC_DEEP=1 VRT_POOL=0
run c3_reseed "--config c3 --steps 10 --warmup 3 --no-cpu --reseed" A=1
run c5_cpu "--config c5 --steps 3 --warmup 1" A=1
