"""What a process that used the GPU costs AFTER its last line of Python: wall of a subprocess minus the time at which it reports being
done, for (a) the HIP runtime started and 40 MB pinned, (b) the bench index open on the device, (c) the index open and 0.7 GB of host
memory pinned (fgpu_prepare_host), (d) as (c) plus the five worker results (fgpu_stream_prepare). All exit through os._exit."""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fulgor_amd import synth
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, _ = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
CODE = r'''
import os, sys, time
sys.path.insert(0, %r)
t0 = float(sys.argv[1]); what = sys.argv[2]; fg = sys.argv[3]
import fulgor_amd
from fulgor_amd import _native
from fulgor_amd.index import prepare_host
L = _native.lib()
if what == "runtime":  # the HIP runtime started on device 0 and ten small buffers pinned (40 MB): as little as the library lets one ask for
    prepare_host(0, reader_threads=1, workers=1, batch=1, text_bytes_per_read=1, out_bytes_per_read=0)
else:
    ix = fulgor_amd.Index(fg, device=0)
    if what in ("pinned", "results"):
        prepare_host(0, out_bytes_per_read=256)
    if what == "results":
        ix.stream_prepare(2, 0, 0, 150, 256)
print("done %%.3f" %% (time.time() - t0), flush=True)
os._exit(0)
''' % ROOT
for what in ("runtime", "index", "pinned", "results"):
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-c", CODE, "%.6f" % t0, what, fg], capture_output=True, text=True, timeout=600)
        wall = time.time() - t0
        done = float(r.stdout.split("done")[1].split()[0]) if "done" in r.stdout else float("nan")
        print("%-8s wall %.3f s, last line of Python at %.3f s, after that %.3f s" % (what, wall, done, wall - done), flush=True)
