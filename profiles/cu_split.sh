#!/bin/bash
# CU partition measurements (round 4): how the kernels of a pass scale with the CUs they may use (FULGOR_CU_RANGE), and the
# pipelined step (bench.py --pipeline 1: lookup of pass t + 1 beside the colour stage of pass t) with and without FULGOR_CU_SPLIT.
mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d['kernels']
    print(sys.argv[1], round(d['value'] / 1e6, 1), 'M reads/s', d['ms_per_step'], 'ms/step', {n: round(v['avg_ms'] * v['launches'] / d['steps'], 3) for n, v in k.items()})
except Exception as e:
    print(sys.argv[1], 'failed', e)
PY
}
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1"
for n in 256 192 160 128 96 64; do
  FULGOR_CU_RANGE=0:$n $B 2>/dev/null | tail -1 > gpurun_out/cu_range_$n.json; show "range 0:$n" gpurun_out/cu_range_$n.json
done
$B --chunk 2500000 2>/dev/null | tail -1 > gpurun_out/cu_plain_c2500k.json; show "one stream, chunk 2.5M" gpurun_out/cu_plain_c2500k.json
$B --pipeline 1 --chunk 2500000 2>/dev/null | tail -1 > gpurun_out/cu_pipe_nosplit.json; show "pipeline, no split, chunk 2.5M" gpurun_out/cu_pipe_nosplit.json
for sp in 80 96 112 128; do
  for c in 2500000 5000000; do
    FULGOR_CU_SPLIT=$sp $B --pipeline 1 --chunk $c 2>/dev/null | tail -1 > gpurun_out/cu_pipe_${sp}_$c.json; show "pipeline, split $sp, chunk $c" gpurun_out/cu_pipe_${sp}_$c.json
  done
done
