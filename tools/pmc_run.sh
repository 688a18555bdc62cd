#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc_run.sh TAG "bench args"   -> gpurun_out/pmc_TAG_{1,2,3}
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; ARGS=$2
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"
P3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --pmc $P --output-format csv -d $O/pmc_${TAG}_$i -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --no-cpu --no-context > $O/pmc_${TAG}_$i.json 2> $O/pmc_${TAG}_$i.err
  i=$((i+1))
done
python3 $R/tools/pmc_summary.py $O/pmc_${TAG}_1 $O/pmc_${TAG}_2 $O/pmc_${TAG}_3 > $O/pmc_${TAG}_summary.txt
