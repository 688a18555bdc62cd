#!/bin/bash
set -o pipefail
O=gpurun_out/r03f; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:12])"
}
run c3_pool1 VRT_POOL=1; run c3_pool0 VRT_POOL=0
for k in "64 1" "56 2" "48 2" "40 3" "32 3"; do set -- $k; run c3_keep$1_it$2 VRT_POOL_KEEP=$1 VRT_POOL_ITERS=$2; done
for th in "40 64" "44 56" "40 56"; do set -- $th; run c3_t$1_$2 VRT_POOL_T_HIT=$1 VRT_POOL_T_END=$2; done
CFG=c5 STEPS=3 WARM=1 run c5_pool0 VRT_POOL=0
for k in "64 1" "56 2" "48 3" "40 3" "32 4" "40 5"; do set -- $k; CFG=c5 STEPS=3 WARM=1 run c5_keep$1_it$2 VRT_POOL_KEEP=$1 VRT_POOL_ITERS=$2; done
VRT_POOL=1 VRT_DIAG=1 timeout -k 10 300 python tools/diag_march.py c3 > $O/diag_c3_pool1.txt 2>&1; grep -v amdgpu.ids $O/diag_c3_pool1.txt | grep "lanes per\|executions\|cycle shares\|ray pool\|wave cycles"
VRT_POOL=1 VRT_DIAG=1 timeout -k 10 400 python tools/diag_march.py c5 > $O/diag_c5_pool1.txt 2>&1; grep -v amdgpu.ids $O/diag_c5_pool1.txt | grep "lanes per\|executions\|cycle shares\|ray pool\|wave cycles"
export VRT_POOL=1; bash tools/pmc_run.sh r03f_c3_pool1 "--config c3" || exit 1
bash tools/pmc_run.sh r03f_c5_pool1 "--config c5" || exit 1
grep -A26 "march_pool_kernel<8, [01]>" gpurun_out/pmc_r03f_c3_pool1_summary.txt gpurun_out/pmc_r03f_c5_pool1_summary.txt | grep "INSTS_VALU \|THREAD_CYCLES\|WAVE_CYCLES\|INSTS_LDS\|INSTS_SALU"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knobs or random_scenes or retrace or third" > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
