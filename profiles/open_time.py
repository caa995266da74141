import glob, os, sys, time
sys.path.insert(0, "/root/repo")
import fulgor_amd
from fulgor_amd import synth
ROOT="/root/repo"
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
for i in range(2):
    t0=time.perf_counter(); ix = fulgor_amd.Index(fg, device=0); t1=time.perf_counter(); print("open %.2f s (%s, %.0f MB)" % (t1-t0, fg, os.path.getsize(fg)/1e6)); ix.close()
