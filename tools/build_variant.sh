#!/bin/bash
# Builds a measurement variant of the HIP library next to the shipped one: tools/build_variant.sh <name> <hipcc flags...>
#   tools/build_variant.sh pool32 -DVRT_POOL_SLOTS=32   ->  python_raytracer_amd/_vrt_pool32.so   (load it with VRT_SO=<path>)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -shared "$@" \
    python_raytracer_amd/csrc/vrt_kernels.hip -o python_raytracer_amd/_vrt_${name}.so
echo python_raytracer_amd/_vrt_${name}.so
