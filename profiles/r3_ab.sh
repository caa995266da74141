#!/bin/bash
# A/B of the round-3 colour-stage changes on one box: dense rows x locality order x small-result bypass (bench.py --rows/--order/--small)
# usage: bash profiles/r3_ab.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "knobs or s4546_full_intersection or compressed_formatter or hit_counts or edge_batches or golden" 2>&1 | tail -5 > gpurun_out/${TAG}_pytest.txt
for cfg in ${CFGS:-0:0:0 1:0:0 1:0:1 1:1:0 1:1:1}; do
  set -- ${cfg//:/ }
  timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 1 --rows $1 --order $2 --small $3 2> gpurun_out/${TAG}_r$1o$2s$3.err | tail -1 > gpurun_out/${TAG}_r$1o$2s$3.json
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/%s_r*o*s*.json" % os.environ.get("TAG", "*"))):
    try:
        j = json.load(open(f))
        print(os.path.basename(f), round(j["value"] / 1e6, 1), "M reads/s", {k: v["avg_ms"] for k, v in j["kernels"].items()}, "stage", j["roofline"]["stage"]["achieved"])
    except Exception as e:
        print(f, "unreadable", e)
PY
cat gpurun_out/${TAG}_pytest.txt
