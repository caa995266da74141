#!/bin/bash
# round 3: (1) FETCH_SIZE / WRITE_SIZE calibration; (2) TCC hit rate + FETCH per launch of the colour kernels with and without
# the locality order. usage: bash profiles/r3_pmc.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C=$R/profiles/micro/fetch_calibration
$C > $OUT/calib_known.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o c -- $C > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o c -- $C > /dev/null 2>&1
python $R/profiles/fetch_calibration.py $OUT/calib_known.txt $OUT/calib_fetch $OUT/calib_write > $OUT/fetch_calibration.txt 2>&1
for o in 0 1; do
  B="python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --order $o --small 0"
  timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/o${o}/pmc_tcc -o t -- $B > /dev/null 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/o${o}/pmc_fetch -o t -- $B > /dev/null 2>&1
  mkdir -p $OUT/o${o}/stats
  python $R/profiles/summarize.py $OUT/o${o} o${o} > $OUT/order${o}_summary.txt 2>&1
done
cat $OUT/fetch_calibration.txt
grep -h "k2a\|k1_lookup\|k2b" $OUT/order0_summary.txt $OUT/order1_summary.txt
