for B in 24 25 26; do echo B=$B; VRT_BATCH_LOG2=$B python tools/exp_share.py 2>&1 | grep "traversed 1" || exit 1; done
for B in 24 26 28; do echo B=$B; VRT_BATCH_LOG2=$B python bench.py --config c5 --steps 3 --warmup 1 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])" || exit 1; done
