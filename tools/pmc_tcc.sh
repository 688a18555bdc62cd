#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc_tcc.sh TAG "bench args"   -> gpurun_out/pmc_TAG_tcc{,2}
# L2 (TCC) counters of the march: hits, misses and the memory-side read requests (64-byte and 32-byte), each pass its own
# rocprofv3 run (four TCC counter slots per pass on gfx950: MI355X_MICROARCH.md, "rocprofv3 PMC slots")
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; ARGS=$2
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/pmc_${TAG}_tcc -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-context > $O/pmc_${TAG}_tcc.json 2> $O/pmc_${TAG}_tcc.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum --output-format csv -d $O/pmc_${TAG}_tcc2 -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-context > $O/pmc_${TAG}_tcc2.json 2> $O/pmc_${TAG}_tcc2.err
python3 $R/tools/pmc_summary.py $O/pmc_${TAG}_tcc $O/pmc_${TAG}_tcc2 > $O/pmc_${TAG}_tcc_summary.txt
