"""One cold `pseudoalign` command on the bench workload: a fresh `python -m fulgor_amd pseudoalign --format compressed -o /dev/null`
process on a FASTQ file of n reads (tmpfs), as a user of the drop-in runs it. Prints the wall clock of the command, the command's own
timeline (FULGOR_CLI_TIMELINE: interpreter start, imports, index open by stage, query) and its `--verbose` summary.
python profiles/cli_cold.py [n reads] [repeats] [extra args of the command ...]"""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fulgor_amd import synth
from fulgor_amd.reads import ReadGenerator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra_args = sys.argv[3:]
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
b, o = ReadGenerator(g, raw_sequences=extra).generate(0, n, 150, 42)
path = "/dev/shm/cold_%d.fq" % os.getpid()
rec = np.empty((n, 316), dtype=np.uint8)
ids = np.arange(n, dtype=np.int64)
rec[:, 0], rec[:, 1], rec[:, 11] = ord("@"), ord("r"), ord("\n")
for d in range(9):
    rec[:, 2 + d] = ord("0") + (ids // 10 ** (8 - d)) % 10
rec[:, 12:162] = np.asarray(b).reshape(n, 150)
rec[:, 162:165] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 165:-1] = ord("I")
rec[:, -1] = ord("\n")
rec.tofile(path)
del rec, b, o
print("index %s (%.2f GB), query %d reads (%.2f GB of FASTQ on tmpfs)" % (os.path.basename(fg), os.path.getsize(fg) / 1e9, n, os.path.getsize(path) / 1e9), flush=True)
try:
    open(fg, "rb").read()  # (the index file in the page cache, as after any earlier command on the same index)
    for rep in range(reps):
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "fulgor_amd", "pseudoalign", "-i", fg, "-q", path, "-o", "/dev/null", "--format", "compressed", "--verbose"] + extra_args,
                           cwd=ROOT, env=dict(os.environ, FULGOR_CLI_TIMELINE="%.6f" % t0, FULGOR_VERBOSE_LOAD="1"), capture_output=True, text=True, timeout=1200)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr[-3000:]
        print("run %d: wall %.3f s for the whole command" % (rep, dt))
        for line in (r.stderr + r.stdout).splitlines():
            if line.strip() and "amdgpu.ids" not in line:
                print("    " + line)
        sys.stdout.flush()
finally:
    if os.path.exists(path):
        os.remove(path)
