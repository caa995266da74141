"""Loop counts of the lookup kernel on the bench workload (instrumented build of the library, -DFG_K1_STATS).
usage (GPU box): python profiles/k1_stats.py [reads]"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "gpurun_out", "libfulgor_gpu_stats.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DFG_K1_STATS",
                    os.path.join(ROOT, "fulgor_amd", "csrc", "fulgor_gpu.hip"), "-o", so, "-lz"], check=True)
from fulgor_amd import _build
_build.LIB_GPU = so  # the instrumented library instead of the product one (this script only)
import fulgor_amd
from fulgor_amd import synth, _native
from fulgor_amd.reads import ReadGenerator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
gen = ReadGenerator(g, raw_sequences=extra)
b, o = gen.generate(0, n, 150, 42)
ix = fulgor_amd.Index(fg, device=0)
reads = ix.upload_reads(b, o)
res = ix.new_result()
lib = _native.lib()
out = (ctypes.c_ulonglong * 16)()
lib.fgpu_debug_k1_stats(out, 1)
ix.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, 0, n)
lib.fgpu_debug_k1_stats(out, 0)
names = ["reads", "passes", "chunks", "batches", "runs", "pairs", "heads", "sum of maxseg", "overflow retries", "heads from pairs", "lanes with one matching record", "lanes with several", "batches with such a lane", "live lanes"]
for i, nm in enumerate(names):
    print("%-18s %12d   per read %.3f   per pass %.3f" % (nm, out[i], out[i] / max(1, out[0]), out[i] / max(1, out[1])))
