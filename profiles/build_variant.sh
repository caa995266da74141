#!/bin/bash
# builds an alternative libfulgor_gpu.so with extra -D flags: bash profiles/build_variant.sh <name> [-DFLAG ...]  ->  build/variants/<name>.so
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread "$@" $R/fulgor_amd/csrc/fulgor_gpu.hip -o $R/build/variants/$name.so -lz -ldl -lhsa-runtime64 2>&1 | grep -v "warning\|^$\|generated" | head -20
ls -la $R/build/variants/$name.so | awk '{print $5, $9}'
