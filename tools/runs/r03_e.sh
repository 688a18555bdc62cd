#!/bin/bash
set -o pipefail
O=gpurun_out/r03e; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:12])"
}
run c3_pool1 VRT_POOL=1; run c3_pool0 VRT_POOL=0
for th in "48 56" "48 64" "44 60" "52 60" "40 64" "48 72"; do set -- $th; run c3_pool1_t$1_$2 VRT_POOL=1 VRT_POOL_T_HIT=$1 VRT_POOL_T_END=$2; done
CFG=c2 run c2_pool1_forced VRT_POOL=1 VRT_POOL_MIN_RAYS=0; CFG=c2 run c2_default
for w in 8 4 2; do
echo "== share 1/$w"; EXP_WORLDS=$w VRT_POOL=1 VRT_POOL_MIN_RAYS=0 timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world; EXP_WORLDS=$w VRT_POOL=0 timeout -k 10 300 python tools/exp_share.py 2>&1 | grep world
done
for th in "48 48" "32 32" "24 24" "40 56" "56 56"; do set -- $th; CFG=c5 STEPS=3 WARM=1 run c5_pool1_t$1_$2 VRT_POOL=1 VRT_POOL_T_HIT=$1 VRT_POOL_T_END=$2; done
CFG=c5 STEPS=3 WARM=1 run c5_pool0 VRT_POOL=0
export VRT_POOL=1; bash tools/pmc_run.sh r03e_c5_pool1 "--config c5" || exit 1
export VRT_POOL=0; bash tools/pmc_run.sh r03e_c5_pool0 "--config c5" || exit 1
grep -A26 "march_pool_kernel<8, 0>\|march_kernel<8, 0, false, false, 0>" gpurun_out/pmc_r03e_c5_pool1_summary.txt gpurun_out/pmc_r03e_c5_pool0_summary.txt | grep "INSTS_VALU \|THREAD_CYCLES\|WAVE_CYCLES\|INSTS_LDS\|INSTS_SALU\|WAIT\|INSTS_SMEM\|ACTIVE_INST_VALU\|BUSY_CYC"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knobs or random_scenes or retrace or third" > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
