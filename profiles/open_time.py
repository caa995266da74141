import glob, os, sys, time
sys.path.insert(0, "/root/repo")
t0 = time.perf_counter()
import fulgor_amd
from fulgor_amd import synth
t1 = time.perf_counter()
g = sorted(glob.glob("/root/repo/tests/data/salmonella_10/*.fasta.gz"))
fg, extra = synth.ensure_s4546("/root/repo/data", g)
print("index file: %.2f GB" % (os.path.getsize(fg) / 1e9), flush=True)
for i in range(3):
    t2 = time.perf_counter()
    ix = fulgor_amd.Index(fg, device=0)
    t3 = time.perf_counter()
    print("open %d: %.2f s (import of the package %.2f s)" % (i, t3 - t2, t1 - t0), flush=True)
    ix.close()
