#!/bin/bash
# tools/ab_reseed.sh OUTDIR: the uncached frame (bench.py --reseed) with the march making its own ray records (default) and
# with raygen_tile_kernel writing a ray table (VRT_FUSE_RAYGEN=0), alternating; one line per run
out=$1; mkdir -p $out
for f in 1 0 1 0; do
  VRT_FUSE_RAYGEN=$f python bench.py --reseed --no-cpu --steps 10 ${AB_FLAGS} > $out/reseed_f$f.json 2> $out/reseed_f$f.err || exit 1
  python - $out/reseed_f$f.json $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fuse", sys.argv[2], d["ms_per_step"], d["kernel_ms_per_step"], d["config"]["image_sha256"][:8])
PY
done
