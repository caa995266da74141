#!/bin/bash
# Which part of k2b_expand conflicts in LDS, and what it costs: the product build against two knock-outs (-DFG_K2B_KO=1: no hit
# counter adds; =2: no stage scatter stores), times from bench.py, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE from rocprofv3 --pmc.
# usage (GPU box): bash profiles/k2b_knockout.sh <tag>   (the variants are built on the CPU side: profiles/build_variant.sh k2bko1 -DFG_K2B_KO=1 ...)
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in product k2bko1 k2bko2; do
  if [ $v = product ]; then unset FULGOR_LIB_GPU; else export FULGOR_LIB_GPU=$R/build/variants/$v.so; fi
  for algo in full-intersection threshold-union; do
    B="python $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --algo $algo"
    timeout 300 $B 2> /dev/null | tail -1 > $OUT/${v}_${algo}.json
    rm -rf $OUT/pmc_${v}_${algo}
    timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_${v}_${algo} -o x -- $B > /dev/null 2>&1
    python - $v $algo $OUT <<'PY'
import csv, glob, json, sys
from collections import defaultdict
v, algo, out = sys.argv[1:4]
acc, n = defaultdict(float), set()
for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, v, algo), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k2b_expand" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
try:
    j = json.load(open("%s/%s_%s.json" % (out, v, algo)))
    ms = j["kernels"]["k2b_expand"]["avg_ms"]
except Exception as e:
    ms = float("nan")
reads = 1e7 * max(1, len(n))
print("%-8s %-18s k2b %.3f ms  launches %d  VALU/read %.1f  LDS/read %.1f  bank conflict cycles / LDS active cycles = %.3f" % (
    v, algo, ms, len(n), acc["SQ_INSTS_VALU"] / reads, acc["SQ_INSTS_LDS"] / reads, acc["SQ_LDS_BANK_CONFLICT"] / max(1.0, acc["SQ_LDS_IDX_ACTIVE"])))
PY
  done
done | tee $OUT/k2b_knockout.txt
