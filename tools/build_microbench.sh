#!/bin/bash
# builds tools/microbench (gfx950) from tools/microbench.hip; the binary travels to the GPU box with gpurun
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 microbench.hip -o microbench
