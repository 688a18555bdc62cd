#!/bin/bash
O=gpurun_out/r03au; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -x --timeout 300 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:8])"
}
CFG=c5 STEPS=3 WARM=1 run c5_ident A=1; CFG=c5 STEPS=3 WARM=1 run c5_read VRT_TABLE_IDENTITY=0; CFG=c5 STEPS=3 WARM=1 run c5_ident_lanes VRT_POOL=0
run c3_now A=1; CFG=c2 STEPS=20 run c2_now A=1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
