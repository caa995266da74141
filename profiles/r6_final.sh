#!/bin/bash
# round 6, final build: rocprofv3 stats + PMC (FI and TU, 10 M reads per launch) and the traffic table from them, the default bench line
# (with its secondary workloads, the PCIe / command-line legs incl. the cold command, and the CPU baseline), salmonella_10, bench.py under
# torchrun with two ranks on the one GPU, the cold command's timeline, the full-size dump round trip, the rates of --deduplicate and the
# k-mer tools, the read-length sweep, and the soak against the oracle.          usage: bash profiles/r6_final.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r6
mkdir -p $O
bash profiles/collect.sh $TAG --steps 2 --warmup 1 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_$TAG $TAG > $O/${TAG}_s4546syn_10M_summary.txt 2>&1
cp gpurun_out/prof_$TAG/stats/${TAG}_kernel_stats.csv $O/${TAG}_s4546syn_10M_kernel_stats.csv
python profiles/traffic_from_pmc.py $O/${TAG}_s4546syn_10M_summary.txt 10000000 > $O/traffic_$TAG.json 2> $O/traffic_$TAG.err
bash profiles/collect.sh ${TAG}tu --steps 2 --warmup 1 --algo threshold-union > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_${TAG}tu ${TAG}tu > $O/${TAG}_s4546syn_tu_10M_summary.txt 2>&1
cp gpurun_out/prof_${TAG}tu/stats/${TAG}tu_kernel_stats.csv $O/${TAG}_s4546syn_tu_10M_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}tu
timeout 1800 python bench.py > $O/bench_s4546syn_10M_$TAG.json 2> $O/bench_fi_$TAG.err
cp gpurun_out/bench_detail.json $O/bench_detail_s4546syn_10M_$TAG.json
timeout 600 python bench.py --workload s10 --no-secondary > $O/bench_s10_1M_$TAG.json 2> $O/bench_s10_$TAG.err
FULGOR_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --workload s10 --reads 300000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_torchrun2_$TAG.json 2> $O/bench_torchrun2_$TAG.err
python profiles/cli_cold.py 10000000 4 2>&1 | grep -v "amdgpu.ids\|^    [0-9]" > $O/cli_cold_$TAG.txt
FULGOR_NO_PREPARE=1 python profiles/cli_cold.py 10000000 3 2>&1 | grep -v "amdgpu.ids\|^    [0-9]" > $O/cli_cold_noprep_$TAG.txt
python profiles/exit_cost.py 2>&1 | grep -v amdgpu.ids > $O/exit_cost_$TAG.txt
python profiles/dump_roundtrip.py --bench 2>&1 | grep -v "amdgpu.ids\|^core\|^total\|^colour stream\|^dictionary" > $O/dump_roundtrip_full_$TAG.txt
python profiles/dedup_and_kmer_tools.py 2>&1 | grep -v "amdgpu.ids\|^core\|^total\|^colour stream\|^dictionary" > $O/dedup_and_kmer_tools_$TAG.txt
python profiles/read_length_sweep.py 2>&1 | grep bases > $O/read_length_sweep_$TAG.txt
python profiles/k3r_mandatory_stats.py 200000 0.8 2>&1 | grep -v "amdgpu.ids\|^core\|^total\|^colour stream\|^dictionary" > $O/k3r_list_histogram_tau08_$TAG.txt
timeout 1500 python profiles/soak_parity.py 10 > $O/soak_parity_10M_$TAG.txt 2>&1
tail -c 400 $O/bench_s4546syn_10M_$TAG.json; echo; wc -c $O/bench_s4546syn_10M_$TAG.json; tail -3 $O/soak_parity_10M_$TAG.txt; grep wall $O/cli_cold_$TAG.txt; tail -4 $O/dump_roundtrip_full_$TAG.txt
