#!/bin/bash
O=gpurun_out/r03l; mkdir -p $O
VRT_POOL=0 python tools/sweep_pool.py c3 -,-,-,-,-,- 2>&1 | tee $O/sweep_c3.txt
python tools/sweep_pool.py c3 -,-,-,-,-,- 40,60,8,8,48,3 36,60,8,8,40,4 44,60,6,8,40,3 2>&1 | tee -a $O/sweep_c3.txt
VRT_POOL=0 python tools/sweep_pool.py c5 -,-,-,-,-,- 2>&1 | tee $O/sweep_c5.txt
python tools/sweep_pool.py c5 -,-,-,-,-,- 48,32,4,8,48,4 44,32,4,8,40,5 2>&1 | tee -a $O/sweep_c5.txt
export VRT_POOL=1; bash tools/pmc_run.sh r03l_c3_pool1 "--config c3" || exit 1
grep -A26 "march_pool_kernel<8, 1>" gpurun_out/pmc_r03l_c3_pool1_summary.txt | grep "INSTS_VALU \|THREAD_CYCLES\|WAVE_CYCLES\|INSTS_LDS\|INSTS_SALU"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knobs or random_scenes or retrace or third or axis" > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
