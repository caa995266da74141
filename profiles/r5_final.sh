#!/bin/bash
# round 5, final build: rocprofv3 stats + PMC (FI and TU, 10 M reads per launch), the default bench line (with its secondary
# workloads, PCIe legs and CPU baseline), salmonella_10, bench.py under torchrun with two ranks on the one GPU, the per-phase
# instruction counts of the lookup kernel, the result-size histogram of the workload, and the soak against the oracle.
# usage: bash profiles/r5_final.sh <tag>     (knock-out variants: profiles/build_variant.sh k1stop{1,2,3} -DFG_K1_STOP={1,2,3})
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
bash profiles/collect.sh $TAG --steps 2 --warmup 1 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_$TAG $TAG > gpurun_out/${TAG}_s4546syn_10M_summary.txt 2>&1
cp gpurun_out/prof_$TAG/stats/${TAG}_kernel_stats.csv gpurun_out/${TAG}_s4546syn_10M_kernel_stats.csv
bash profiles/collect.sh ${TAG}tu --steps 2 --warmup 1 --algo threshold-union > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_${TAG}tu ${TAG}tu > gpurun_out/${TAG}_s4546syn_tu_10M_summary.txt 2>&1
cp gpurun_out/prof_${TAG}tu/stats/${TAG}tu_kernel_stats.csv gpurun_out/${TAG}_s4546syn_tu_10M_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG gpurun_out/prof_${TAG}tu
timeout 1800 python bench.py > gpurun_out/bench_s4546syn_10M_$TAG.json 2> gpurun_out/bench_fi_$TAG.err
timeout 600 python bench.py --workload s10 --no-secondary > gpurun_out/bench_s10_1M_$TAG.json 2> gpurun_out/bench_s10_$TAG.err
FULGOR_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --workload s10 --reads 300000 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun2_$TAG.json 2> gpurun_out/bench_torchrun2_$TAG.err
python profiles/k1_variant_time.py 10000000 build/variants/k1stop1.so,build/variants/k1stop2.so,build/variants/k1stop3.so,fulgor_amd/libfulgor_gpu.so 2>&1 | grep k1_lookup > gpurun_out/k1_phase_times_$TAG.txt
bash profiles/k1_phase_counts.sh build/variants/k1stop1.so build/variants/k1stop2.so build/variants/k1stop3.so fulgor_amd/libfulgor_gpu.so 2>&1 | grep launches >> gpurun_out/k1_phase_times_$TAG.txt
python - > gpurun_out/result_sizes_$TAG.txt 2>&1 <<'PY'
import numpy as np, bench, fulgor_amd
fg, gen, desc = bench.prepare_workload("s4546syn")
ix = fulgor_amd.Index(fg, device=0)
b, o = gen.generate(0, 1000000, 150, 42)
for name, (off, col) in (("full intersection", ix.pseudoalign_full_intersection_batch(b, o)), ("threshold union 0.8", ix.pseudoalign_threshold_union_batch(b, o, 0.8))):
    sz = np.diff(off.astype(np.int64))
    edges = [0, 1, 17, 33, 65, 129, 513, 1137, 2049, 3410, 4547]
    h = np.histogram(sz, bins=edges)[0]
    print(name, "colours per read: mean %.1f" % sz.mean(), " ".join("[%d,%d):%.1f%%" % (edges[i], edges[i + 1], 100.0 * h[i] / len(sz)) for i in range(len(h))),
          "| share of the colours written:", " ".join("%.1f%%" % (100.0 * sz[(sz >= edges[i]) & (sz < edges[i + 1])].sum() / max(1, sz.sum())) for i in range(len(h))))
PY
timeout 1500 python profiles/soak_parity.py 10 > gpurun_out/soak_parity_10M_$TAG.txt 2>&1
tail -c 300 gpurun_out/bench_torchrun2_$TAG.json; tail -3 gpurun_out/soak_parity_10M_$TAG.txt; cat gpurun_out/k1_phase_times_$TAG.txt gpurun_out/result_sizes_$TAG.txt
# round 5: the streamed command-line path under the kernel trace (copies on kernel-free streams; rocprofv3 shows them as __amd_rocclr_copyBuffer
# dispatches of the copy queues) and its own timeline without the profiler
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}e2e -o ${TAG}e2e -- python $R/profiles/e2e_once.py 10000000 24 5 262144 4 > $R/gpurun_out/e2e_once_rocprof_$TAG.txt 2>&1
cd $R
cp gpurun_out/prof_${TAG}e2e/${TAG}e2e_kernel_stats.csv gpurun_out/${TAG}_e2e_stream_10M_kernel_stats.csv 2>/dev/null || find gpurun_out/prof_${TAG}e2e -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_e2e_stream_10M_kernel_stats.csv \;
rm -rf gpurun_out/prof_${TAG}e2e
E2E_COLD=1 timeout 600 python profiles/e2e_once.py 10000000 24 5 262144 6 2>&1 | grep -v "^W2026\|^E2026" > gpurun_out/e2e_once_$TAG.txt
grep "^run" gpurun_out/e2e_once_$TAG.txt
