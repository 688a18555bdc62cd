#!/bin/bash
# usage (via gpurun, from the repo root): bash tools/lookup_variants.sh
# rocprofv3 kernel statistics of the measurement variants of march_kernel at BASELINE config 5 and config 3:
#   VRT_LOOKUP = 0 material bytes (shipped), 1 occupancy words in registers, 2 8^3 occupancy bricks staged in LDS
#   VRT_ROLES  = 1 wave roles: one loader / finisher wave per workgroup, three marching waves
# all with 4-step speculation (the lookup variants spill registers with 8 steps) -> gpurun_out/lk_<cfg>_<variant>/ ;
# tools/save_profiles.py copies the march rows into profiles/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
export VRT_SPEC_DEEP=0
for cfg in c5 c3; do
  steps=5; [ $cfg = c5 ] && steps=3
  for lk in 0 1 2; do
    export VRT_LOOKUP=$lk
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/lk_${cfg}_$lk -- python3 $R/bench.py --config $cfg --steps $steps --warmup 1 --no-cpu --no-context > $O/lk_${cfg}_$lk.json
  done
  export VRT_LOOKUP=0 VRT_ROLES=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/lk_${cfg}_roles -- python3 $R/bench.py --config $cfg --steps $steps --warmup 1 --no-cpu --no-context > $O/lk_${cfg}_roles.json
  unset VRT_ROLES
done
echo done
