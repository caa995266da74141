#!/bin/bash
# kernel times of variant builds (profiles/build_variant.sh) on the bench workload: bash profiles/variant_bench.sh "<bench.py arguments>" <name> [<name> ...]
# ("shipped" = the library in the tree)
R=$(cd "$(dirname "$0")/.." && pwd)
args=$1; shift
for v in "$@"; do
  if [ "$v" = shipped ]; then lib=""; else lib=$R/build_r6/$v.so; fi
  FULGOR_LIB_GPU=$lib python $R/bench.py $args --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print('%-16s %8.1f M reads/s  %7.3f ms/step  %s' % ('$v', l['value']/1e6, l['ms_per_step'], ' '.join('%s %.3f' % (k, v) for k, v in l['kernels_ms'].items())))"
done
