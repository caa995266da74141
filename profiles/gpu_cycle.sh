timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 280 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_cur.json
timeout 200 python bench.py --no-cpu-baseline --algo threshold-union --reads 2000000 2>&1 | tail -1 > gpurun_out/bench_cur_tu.json
