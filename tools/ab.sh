#!/bin/bash
# usage (via gpurun, from the repo root): bash tools/ab.sh OUTDIR VARIANT...
# A/B of library builds on one box: config 3 (ray pool) and config 5 with one ray per lane, for every variant
# (python_raytracer_amd/_vrt_VARIANT.so from tools/build_variant.sh; "base" = the shipped library).  AB_FLAGS: extra bench flags.
out=$1; shift
mkdir -p gpurun_out/$out && cd gpurun_out/$out
for v in "$@"; do
  so=""; [ $v != base ] && so=$GRAFT_REPO_ROOT/python_raytracer_amd/_vrt_$v.so
  VRT_SO=$so timeout -k 10 200 python ../../bench.py --no-cpu --steps 20 $AB_FLAGS > c3_$v.json 2> c3_$v.err
  [ -z "$AB_SKIP_C5" ] && VRT_SO=$so VRT_POOL=${AB_C5_POOL:-0} timeout -k 10 300 python ../../bench.py --no-cpu --config c5 --steps 3 --warmup 1 $AB_FLAGS > c5_$v.json 2> c5_$v.err
done
for f in *.json; do python -c "
import json,sys
d=json.load(open('$f'))
print('$f', d['ms_per_step'], d['kernel_ms_per_step']['march'], d['config']['image_sha256'][:12])"; done
