#!/bin/bash
# builds an alternative libfulgor_gpu.so with extra -D flags: bash profiles/build_variant.sh <name> [-DFLAG ...]  ->  build_r6/<name>.so
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build_r6
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread "$@" $R/fulgor_amd/csrc/fulgor_gpu.hip -o $R/build_r6/$name.so -lz -ldl -lhsa-runtime64 2>&1 | grep -v "warning\|^$\|generated" | head -20
ls -la $R/build_r6/$name.so | awk '{print $5, $9}'
