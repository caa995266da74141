for st in 1 2 3; do
  timeout 600 python bench.py --reads 4000000 --steps 3 --warmup 1 --no-cpu-baseline --streams $st 2>/dev/null > gpurun_out/st_$st.json
  python - <<PY
import json
d=json.load(open("gpurun_out/st_$st.json"))
print($st, d["value"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
