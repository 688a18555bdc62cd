#!/bin/bash
# usage (via gpurun, from the repo root): bash tools/profile_all.sh
# Writes gpurun_out/{bench_cX.json, bench_cX_reseed.json, prof_cX/, prof_cX_bench.json, pmc_fetch_cX/, pmc_write_cX/};
# afterwards run `python tools/save_profiles.py TAG` in the container to copy the summaries into profiles/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --config c3 > $O/bench_c3.json
python3 $R/bench.py --config c2 --no-cpu > $O/bench_c2.json
python3 $R/bench.py --config c5 --steps 5 --warmup 1 --no-cpu > $O/bench_c5.json
python3 $R/bench.py --config c3 --no-cpu --reseed > $O/bench_c3_reseed.json
python3 $R/bench.py --config c2 --no-cpu --reseed > $O/bench_c2_reseed.json
for cfg in c3 c5 c2; do
  # --no-context: every march launch the profiler sees is the warm-up frame or one of the timed frames
  steps=20; [ $cfg = c5 ] && steps=5
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python3 $R/bench.py --config $cfg --steps $steps --warmup 1 --no-cpu --no-context > $O/prof_${cfg}_bench.json
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$cfg -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --no-context > /dev/null
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$cfg -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --no-context > /dev/null
done
echo done
