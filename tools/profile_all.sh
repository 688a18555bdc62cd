#!/bin/bash
# usage (via gpurun, from the repo root): bash tools/profile_all.sh TAG [1|2]   (two gpurun calls: part 1, then part 2 --
#    together they run longer than one call's limit; without a part number both run)
# 1. the HBM counter passes (FETCH_SIZE, WRITE_SIZE: their own rocprofv3 runs) -> gpurun_out/pmc_{fetch,write}_cX/ and,
#    derived on the spot, profiles/pmc_cX.json (tools/save_profiles.py TAG --pmc-only);
# 2. the bench lines (gpurun_out/bench_cX.json, bench_cX_reseed.json), whose roofline.traffic is read from that file;
# 3. rocprofv3 kernel statistics of the same bench command (gpurun_out/prof_cX/, prof_cX_bench.json).
# Afterwards run `python tools/save_profiles.py TAG` in the container: it copies the summaries into profiles/ and derives
# the same pmc_cX.json from the same counter files.
set -e
TAG=${1:-untagged}; PART=${2:-all}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
if [ $PART != 2 ]; then
# --no-context: every march launch the profiler sees is the warm-up frame or one of the timed frames
for cfg in c3 c5 c2; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$cfg -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --no-context > /dev/null
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$cfg -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --no-context > /dev/null
done
python3 $R/tools/save_profiles.py $TAG --pmc-only
python3 $R/bench.py --config c3 > $O/bench_c3.json
python3 $R/bench.py --config c2 --no-cpu > $O/bench_c2.json
python3 $R/bench.py --config c5 --steps 5 --warmup 1 > $O/bench_c5.json
python3 $R/bench.py --config c3 --no-cpu --reseed > $O/bench_c3_reseed.json
python3 $R/bench.py --config c2 --no-cpu --reseed > $O/bench_c2_reseed.json
# the uncached frame with a ray table written and read back (VRT_FUSE_RAYGEN=0: raygen_tile_kernel), beside the default
VRT_FUSE_RAYGEN=0 python3 $R/bench.py --config c3 --no-cpu --reseed > $O/bench_c3_reseed_table.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_reseed -- python3 $R/bench.py --config c3 --steps 10 --warmup 1 --no-cpu --no-context --reseed > $O/prof_c3_reseed_bench.json
for cfg in c3 c5 c2; do
  steps=20; [ $cfg = c5 ] && steps=5
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$cfg -- python3 $R/bench.py --config $cfg --steps $steps --warmup 1 --no-cpu --no-context > $O/prof_${cfg}_bench.json
done
# the reference's own flow: world resident, chunks + LOD selected on the device every frame (bench.py --world-flow)
python3 $R/bench.py --config c3 --no-cpu --world-flow > $O/bench_c3_world.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_world -- python3 $R/bench.py --config c3 --steps 20 --warmup 1 --no-cpu --no-context --world-flow > $O/prof_c3_world_bench.json
echo part 1 done
fi
[ $PART = 1 ] && exit 0
# ---- part 2: variants of the march on one box, SQ / L2 counters, loop statistics, shares
# the shipped march first, so that every row of the variant table comes from this box
if [ $PART = 2 ]; then
  for cfg in c3 c5; do
    steps=20; [ $cfg = c5 ] && steps=5
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2_$cfg -- python3 $R/bench.py --config $cfg --steps $steps --warmup 1 --no-cpu --no-context > $O/prof2_${cfg}_bench.json
  done
fi
# the one-ray-per-lane march on the same box (VRT_POOL=0), for the pool's variant table
for cfg in c3 c5; do
  steps=20; [ $cfg = c5 ] && steps=5
  VRT_POOL=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${cfg}_lanes -- python3 $R/bench.py --config $cfg --steps $steps --warmup 1 --no-cpu --no-context > $O/prof_${cfg}_lanes_bench.json
done
# the measured variant whose look-ahead crosses chunk borders (VRT_WADDR=1, march_step_w): config 3 with the ray pool,
# config 5 with one ray per lane (its world-axis tables and the ray pools do not fit a workgroup's LDS together)
VRT_WADDR=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_ahead -- python3 $R/bench.py --config c3 --steps 20 --warmup 1 --no-cpu --no-context > $O/prof_c3_ahead_bench.json
VRT_WADDR=1 VRT_POOL=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5_lanes_ahead -- python3 $R/bench.py --config c5 --steps 5 --warmup 1 --no-cpu --no-context > $O/prof_c5_lanes_ahead_bench.json
# SQ counters of both configurations (three passes each), L2 counters, the instrumented build's loop statistics, 1/N shares
cd $R
for cfg in c3 c5; do bash tools/pmc_run.sh $cfg "--config $cfg"; done
for cfg in c3 c5; do bash tools/pmc_tcc.sh $cfg "--config $cfg"; done
VRT_POOL=0 bash tools/pmc_run.sh c3_lanes "--config c3"
for cfg in c3 c5; do VRT_DIAG=1 python3 tools/diag_march.py $cfg 2>&1 | grep -v amdgpu.ids > $O/diag_$cfg.txt; done
for cfg in c3 c5; do VRT_DIAG=2 python3 tools/diag_march.py $cfg 2>&1 | grep -v amdgpu.ids > $O/diag_${cfg}_hist.txt; done
VRT_DIAG=1 DIAG_RESEED=1 python3 tools/diag_march.py c3 2>&1 | grep -v amdgpu.ids > $O/diag_c3_reseed.txt
VRT_WADDR=1 VRT_DIAG=2 python3 tools/diag_march.py c3 2>&1 | grep -v amdgpu.ids > $O/diag_c3_ahead_hist.txt
for w in 0 1; do VRT_WADDR=$w VRT_POOL=0 VRT_DIAG=2 python3 tools/diag_march.py c5 2>&1 | grep -v amdgpu.ids > $O/diag_c5_lanes_w${w}_hist.txt; done
EXP_WORLDS=8,4,2,1 python3 tools/exp_share.py 2>&1 | grep world > $O/share.txt
python3 tools/policy_check.py > $O/policy_check.md
echo done
