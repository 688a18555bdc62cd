#!/bin/bash
# first GPU run of round 3: smoke, pool vs one-ray-per-lane march at c3 / c2 / c5, diag, knob + random-scene parity
set -o pipefail
O=gpurun_out/r03a; mkdir -p $O
export VRT_POOL_VERBOSE=1
echo "== smoke"; timeout -k 10 400 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
for cfg in c3 c2; do
  for pool in 1 0; do
    echo "== bench $cfg pool=$pool"
    VRT_POOL=$pool timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu --no-context > $O/bench_${cfg}_pool$pool.json 2> $O/bench_${cfg}_pool$pool.err || { tail -20 $O/bench_${cfg}_pool$pool.err; exit 1; }
    python - <<PY
import json; d=json.load(open("$O/bench_${cfg}_pool$pool.json")); print("$cfg pool=$pool", d["ms_per_step"], d["kernel_ms_per_step"], d["config"]["image_sha256"][:16])
PY
  done
done
for n in 32 40; do
  echo "== bench c3 pool slots=$n"
  VRT_SO=$PWD/python_raytracer_amd/_vrt_pool$n.so timeout -k 10 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu --no-context > $O/bench_c3_slots$n.json 2> $O/bench_c3_slots$n.err || { tail -20 $O/bench_c3_slots$n.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_c3_slots$n.json')); print('c3 slots $n', d['ms_per_step'], d['kernel_ms_per_step'])"
done
for th in "32 32" "40 40" "56 56" "64 48" "48 32"; do
  set -- $th
  echo "== bench c3 pool t_hit=$1 t_end=$2"
  VRT_POOL_T_HIT=$1 VRT_POOL_T_END=$2 timeout -k 10 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu --no-context > $O/bench_c3_t$1_$2.json 2> $O/bench_c3_t$1_$2.err || { tail -20 $O/bench_c3_t$1_$2.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_c3_t$1_$2.json')); print('c3 t $1 $2', d['ms_per_step'], d['kernel_ms_per_step'])"
done
for pool in 1 0; do
  echo "== diag c3 pool=$pool"
  VRT_POOL=$pool VRT_DIAG=1 timeout -k 10 300 python tools/diag_march.py c3 > $O/diag_c3_pool$pool.txt 2>&1 || { tail -20 $O/diag_c3_pool$pool.txt; exit 1; }
  cat $O/diag_c3_pool$pool.txt
done
for pool in 1 0; do
  echo "== bench c5 pool=$pool"
  VRT_POOL=$pool timeout -k 10 400 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu --no-context > $O/bench_c5_pool$pool.json 2> $O/bench_c5_pool$pool.err || { tail -20 $O/bench_c5_pool$pool.err; exit 1; }
  python -c "import json; d=json.load(open('$O/bench_c5_pool$pool.json')); print('c5 pool=$pool', d['ms_per_step'], d['kernel_ms_per_step'])"
done
echo "== parity subset"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knobs or random_scenes or bit_exact or retrace or third or edge" > $O/pytest_subset.log 2>&1; tail -5 $O/pytest_subset.log
