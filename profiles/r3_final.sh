#!/bin/bash
# round 3, final build: rocprofv3 stats + PMC (FI and TU, 10 M reads per launch), the default bench line (with its secondary
# workloads, PCIe legs and CPU baseline), salmonella_10, and bench.py under torchrun with two ranks on the one GPU.
# usage: bash profiles/r3_final.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash profiles/collect.sh $TAG --steps 2 --warmup 1 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_$TAG $TAG > gpurun_out/${TAG}_s4546syn_10M_summary.txt 2>&1
cp gpurun_out/prof_$TAG/stats/${TAG}_kernel_stats.csv gpurun_out/${TAG}_s4546syn_10M_kernel_stats.csv
bash profiles/collect.sh ${TAG}tu --steps 2 --warmup 1 --algo threshold-union > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_${TAG}tu ${TAG}tu > gpurun_out/${TAG}_s4546syn_tu_10M_summary.txt 2>&1
cp gpurun_out/prof_${TAG}tu/stats/${TAG}tu_kernel_stats.csv gpurun_out/${TAG}_s4546syn_tu_10M_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}tu
timeout 1500 python bench.py > gpurun_out/bench_s4546syn_10M_$TAG.json 2> gpurun_out/bench_fi_$TAG.err
timeout 600 python bench.py --workload s10 --no-secondary > gpurun_out/bench_s10_1M_$TAG.json 2> gpurun_out/bench_s10_$TAG.err
FULGOR_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --workload s10 --reads 300000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun2_$TAG.json 2> gpurun_out/bench_torchrun2_$TAG.err
tail -c 400 gpurun_out/bench_torchrun2_$TAG.json; tail -3 gpurun_out/bench_torchrun2_$TAG.err
