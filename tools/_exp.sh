python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
EXP_WORLDS=32,8,1 python tools/exp_share.py 2>&1 | grep "traversed 1" || exit 1
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu
python bench.py --config c2 --no-cpu
