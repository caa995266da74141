#!/bin/bash
# Full evidence set for one build (run on the GPU box): rocprofv3 stats + PMC at 10M reads per launch (= bench.py's pass size),
# then the bench lines of BASELINE configs[1..3] (and the meta-diff codec) at full size.
# usage: bash profiles/round_evidence.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash profiles/collect.sh $TAG --steps 2 --warmup 1 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_$TAG $TAG > gpurun_out/${TAG}_s4546syn_10M_summary.txt 2>&1
cp gpurun_out/prof_$TAG/stats/${TAG}_kernel_stats.csv gpurun_out/${TAG}_s4546syn_10M_kernel_stats.csv
timeout 900 python bench.py > gpurun_out/bench_s4546syn_10M_$TAG.json 2> gpurun_out/bench_fi_$TAG.err
timeout 900 python bench.py --algo threshold-union > gpurun_out/bench_s4546syn_tu_10M_$TAG.json 2> gpurun_out/bench_tu_$TAG.err
timeout 600 python bench.py --workload s10 > gpurun_out/bench_s10_1M_$TAG.json 2> gpurun_out/bench_s10_$TAG.err
timeout 900 python bench.py --index-type meta-diff --reads 5000000 --no-cpu-baseline > gpurun_out/bench_s4546syn_metadiff_5M_$TAG.json 2> gpurun_out/bench_md_$TAG.err
tail -c 600 gpurun_out/bench_s4546syn_10M_$TAG.json
