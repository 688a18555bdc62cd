#!/bin/bash
set -o pipefail
O=gpurun_out/r03d; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --config ${CFG:-c3} --steps ${STEPS:-10} --warmup ${WARM:-3} --no-cpu --no-context > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:12])"
}
run c3_pool1 VRT_POOL=1; run c3_pool0 VRT_POOL=0
run c3_pool1_fm1 VRT_POOL=1 VRT_SO=$PWD/python_raytracer_amd/_vrt_fm1.so; run c3_pool0_fm1 VRT_POOL=0 VRT_SO=$PWD/python_raytracer_amd/_vrt_fm1.so
run c3_pool0_w5 VRT_POOL=0 VRT_SO=$PWD/python_raytracer_amd/_vrt_w5.so
run c3_pool1_s40 VRT_POOL=1 VRT_SO=$PWD/python_raytracer_amd/_vrt_pool40.so
for th in "40 40" "56 48" "48 56" "64 64"; do set -- $th; run c3_pool1_t$1_$2 VRT_POOL=1 VRT_POOL_T_HIT=$1 VRT_POOL_T_END=$2; done
for sw in "1 1" "8 8" "12 4" "2 16"; do set -- $sw; run c3_pool1_sw$1_$2 VRT_POOL=1 VRT_POOL_SWAP_MIN=$1 VRT_POOL_REFILL_MIN=$2; done
CFG=c2 run c2_pool1 VRT_POOL=1; CFG=c2 run c2_pool0 VRT_POOL=0
CFG=c5 STEPS=3 WARM=1 run c5_pool1 VRT_POOL=1; CFG=c5 STEPS=3 WARM=1 run c5_pool0 VRT_POOL=0
CFG=c5 STEPS=3 WARM=1 run c5_pool0_w5 VRT_POOL=0 VRT_SO=$PWD/python_raytracer_amd/_vrt_w5.so
VRT_POOL=1 VRT_DIAG=1 timeout -k 10 300 python tools/diag_march.py c3 > $O/diag_c3_pool1.txt 2>&1; grep -v amdgpu.ids $O/diag_c3_pool1.txt
VRT_POOL=1 VRT_DIAG=1 timeout -k 10 400 python tools/diag_march.py c5 > $O/diag_c5_pool1.txt 2>&1; grep -v amdgpu.ids $O/diag_c5_pool1.txt
export VRT_POOL=1; bash tools/pmc_run.sh r03d_c3_pool1 "--config c3" || exit 1
grep -A26 "march_pool_kernel<8, 1>" gpurun_out/pmc_r03d_c3_pool1_summary.txt | grep "INSTS_VALU \|THREAD_CYCLES\|WAVE_CYCLES\|INSTS_LDS\|INSTS_SALU\|WAIT\|INSTS_SMEM"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knobs or random_scenes or retrace or third" > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
