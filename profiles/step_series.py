"""Per-pass kernel times of consecutive passes in a fresh process (does the expansion kernel need warming up?).
usage (GPU box): python profiles/step_series.py [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, fulgor_amd
n_pass = int(sys.argv[1]) if len(sys.argv) > 1 else 30
fg, gen, desc = bench.prepare_workload("s4546syn")
ix = fulgor_amd.Index(fg, device=0)
b, o = gen.generate(0, 10000000, 150, 42)
reads = ix.upload_reads(b, o)
res = ix.new_result()
ix.timing_enable(True)
prev = {}
for i in range(n_pass):
    t0 = time.perf_counter()
    ix.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, 0, 10000000)
    wall = (time.perf_counter() - t0) * 1e3
    t = ix.timing()
    cur = {k: v[0] for k, v in t.items()}
    print("pass %2d wall %7.2f ms " % (i, wall) + " ".join("%s %.3f" % (k, cur[k] - prev.get(k, 0.0)) for k in ("k1_lookup", "k2_intersect", "k2b_expand")))
    prev = cur
