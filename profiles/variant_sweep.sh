#!/bin/bash
# the read-length sweep of the lookup kernel for several builds: bash profiles/variant_sweep.sh shipped <name> ...
R=$(cd "$(dirname "$0")/.." && pwd)
for v in "$@"; do
  if [ "$v" = shipped ]; then lib=""; else lib=$R/build_r6/$v.so; fi
  echo "== $v"
  FULGOR_LIB_GPU=$lib python $R/profiles/read_length_sweep.py 2>/dev/null | grep bases
done
