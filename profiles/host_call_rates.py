"""What the host-buffer convenience calls (fgpu_full_intersection / fgpu_threshold_union: the calls of INTEGRATION.md's worker stub) cost per chunk of
reads, against the same chunk through a kept result (upload + run + download).   usage (GPU box): python profiles/host_call_rates.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, fulgor_amd
fg, gen, desc = bench.prepare_workload("s4546syn")
ix = fulgor_amd.Index(fg, device=0)
for n in (1000, 10000, 100000, 1000000):
    b, o = gen.generate(1000, n, 150, 3)
    reps = max(3, min(50, 2000000 // n))
    ix.pseudoalign_full_intersection_batch(b, o)
    t0 = time.perf_counter()
    for _ in range(reps):
        go, gc = ix.pseudoalign_full_intersection_batch(b, o)
    t_call = (time.perf_counter() - t0) / reps
    res = ix.new_result()
    rd = ix.upload_reads(b, o); ix.run(rd, res, fulgor_amd.FULL_INTERSECTION, 0.0); res.download(); rd.close()
    t0 = time.perf_counter()
    for _ in range(reps):
        rd = ix.upload_reads(b, o)
        ix.run(rd, res, fulgor_amd.FULL_INTERSECTION, 0.0)
        go2, gc2 = res.download()
        rd.close()
    t_kept = (time.perf_counter() - t0) / reps
    res.close()
    assert np.array_equal(go, go2) and np.array_equal(gc, gc2)
    print("chunks of %7d reads: fgpu_full_intersection %8.3f ms per call = %6.2f M reads/s;  kept result %8.3f ms = %6.2f M reads/s" % (
        n, t_call * 1e3, n / t_call / 1e6, t_kept * 1e3, n / t_kept / 1e6), flush=True)
