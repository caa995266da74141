"""The compressed device formatter kernel by kernel: one 2^19-read full-intersection pass on the bench index, then
fgpu_result_format_view(compressed) several times. Run under `rocprofv3 --kernel-trace --stats` for the per-kernel times;
prints the mix of record kinds. python profiles/cfmt_split.py [reads] [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, fulgor_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
fg, gen, desc = bench.prepare_workload("s4546syn")
ix = fulgor_amd.Index(fg, device=0)
b, o = gen.generate(0, n, 150, 42)
rd, res = ix.upload_reads(b, o), ix.new_result()
ix.run(rd, res, fulgor_amd.FULL_INTERSECTION)
ix.timing_enable(True)
res.format_view(2, 0)
ix.timing_reset()
t0 = time.perf_counter()
for _ in range(reps):
    out = res.format_view(2, 0)
dt = (time.perf_counter() - t0) / reps
tm = ix.timing()
print("%d reads -> %.1f MB of compressed records; per call: %.3f ms wall, format kernels %.3f ms, scan %.3f ms, D2H %.3f ms"
      % (n, len(out) / 1e6, dt * 1e3, tm["k_format"][0] / reps, tm["scan"][0] / reps, tm["d2h"][0] / reps))
offs, cols = res.download()
sz = np.diff(offs.astype(np.int64))
nc = ix.num_colors()
sp, de = int(0.25 * nc), int(0.75 * nc)
kinds = {"empty": (sz == 0).sum(), "1..16": ((sz > 0) & (sz <= 16)).sum(), "17..sparse<%d" % sp: ((sz > 16) & (sz < sp)).sum(),
         "bitmap": ((sz >= sp) & (sz < de)).sum(), "dense>=%d" % de: (sz >= de).sum()}
print("record kinds: " + ", ".join("%s %.1f%%" % (k_, 100.0 * v / n) for k_, v in kinds.items()))
print("set bits walked by the gap kinds per read (sparse: colours, dense: missing colours): sparse mean %.0f, dense mean %.0f"
      % (sz[(sz > 16) & (sz < sp)].mean() if kinds["17..sparse<%d" % sp] else 0, (nc - sz[sz >= de]).mean() if kinds["dense>=%d" % de] else 0))
