"""Times the kernels of alternative builds of the library on the bench workload (one pass of n reads).
usage (GPU box): python profiles/k1_variant_time.py <n reads> <lib.so> [<lib.so> ...]   (each in a fresh process)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 3 or (len(sys.argv) == 3 and not sys.argv[2].endswith(".so")):
    raise SystemExit(__doc__)
n = int(sys.argv[1])
if len(sys.argv) == 2 or "," in sys.argv[2]:
    for so in sys.argv[2].split(","):
        subprocess.run([sys.executable, __file__, str(n), so])
    raise SystemExit(0)
so = os.path.abspath(sys.argv[2])
from fulgor_amd import _build
_build.LIB_GPU = so
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import ReadGenerator
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
gen = ReadGenerator(g, raw_sequences=extra)
b, o = gen.generate(0, n, 150, 42)
ix = fulgor_amd.Index(fg, device=0)
reads = ix.upload_reads(b, o)
res = ix.new_result()
for _ in range(2):
    ix.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, 0, n)
ix.timing_enable(True)
for _ in range(5):
    ix.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, 0, n)
t = ix.timing()
print(os.path.basename(so), " ".join("%s %.3f ms" % (k, v[0] / v[1]) for k, v in t.items() if v[1]))
