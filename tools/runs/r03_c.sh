#!/bin/bash
set -o pipefail
O=gpurun_out/r03c; mkdir -p $O
for cfg in c3 c2; do
  for pool in 1 0; do
    VRT_POOL=$pool timeout -k 10 300 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu --no-context > $O/bench_${cfg}_pool$pool.json 2> $O/bench_${cfg}_pool$pool.err || { tail -20 $O/bench_${cfg}_pool$pool.err; exit 1; }
    python -c "import json; d=json.load(open('$O/bench_${cfg}_pool$pool.json')); print('$cfg pool=$pool', d['ms_per_step'], d['kernel_ms_per_step'], d['config']['image_sha256'][:16])"
  done
done
for th in "40 40" "40 32" "48 24" "56 40"; do
  set -- $th
  VRT_POOL_T_HIT=$1 VRT_POOL_T_END=$2 timeout -k 10 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu --no-context > $O/bench_c3_t$1_$2.json 2> $O/bench_c3_t$1_$2.err || exit 1
  python -c "import json; d=json.load(open('$O/bench_c3_t$1_$2.json')); print('c3 t $1 $2', d['ms_per_step'], d['kernel_ms_per_step'])"
done
for sw in "1 1" "8 8" "16 16" "4 32"; do
  set -- $sw
  VRT_POOL_SWAP_MIN=$1 VRT_POOL_REFILL_MIN=$2 timeout -k 10 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu --no-context > $O/bench_c3_sw$1_$2.json 2> $O/bench_c3_sw$1_$2.err || exit 1
  python -c "import json; d=json.load(open('$O/bench_c3_sw$1_$2.json')); print('c3 swap_min $1 refill_min $2', d['ms_per_step'], d['kernel_ms_per_step'])"
done
for pool in 1 0; do
  VRT_POOL=$pool timeout -k 10 400 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu --no-context > $O/bench_c5_pool$pool.json 2> $O/bench_c5_pool$pool.err || exit 1
  python -c "import json; d=json.load(open('$O/bench_c5_pool$pool.json')); print('c5 pool=$pool', d['ms_per_step'], d['kernel_ms_per_step'])"
done
VRT_POOL=1 VRT_DIAG=1 timeout -k 10 300 python tools/diag_march.py c3 > $O/diag_c3_pool1.txt 2>&1; grep -v amdgpu.ids $O/diag_c3_pool1.txt
export VRT_POOL=1; bash tools/pmc_run.sh r03c_c3_pool1 "--config c3" || exit 1
export VRT_POOL=0; bash tools/pmc_run.sh r03c_c3_pool0 "--config c3" || exit 1
grep -A26 "march_pool_kernel<8, 1>\|march_kernel<8, 1, false, false, 0>" gpurun_out/pmc_r03c_c3_pool1_summary.txt gpurun_out/pmc_r03c_c3_pool0_summary.txt | grep "INSTS_VALU \|THREAD_CYCLES\|WAVE_CYCLES\|INSTS_LDS\|INSTS_SALU\|WAIT\|INSTS_SMEM"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knobs or random_scenes or retrace or third" > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
