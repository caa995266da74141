#!/bin/bash
# Collects the rocprofv3 evidence for one bench configuration on the GPU box.
# usage: profiles/collect.sh <tag> <bench args...>     (writes gpurun_out/prof_<tag>/...)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-secondary $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $BENCH > $OUT/stats.log 2>&1
pmc() { # name counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o $TAG -- $BENCH > $OUT/pmc_$name.log 2>&1
}
pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR
pmc tcc TCC_HIT_sum TCC_MISS_sum
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc grbm GRBM_GUI_ACTIVE
find $OUT -name "*.csv" | head -50
