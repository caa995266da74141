"""`pseudoalign --gpus N` against `--gpus 1` on a FASTQ file of n reads of the bench workload (both ranks on the one GPU of the box:
FULGOR_SHARE_GPU=1): the output files must be the same bytes for ascii and binary records and parse to the same lists for the
compressed ones (their blocks follow the batches); wall times of the commands. python profiles/cli_ranks_check.py [n reads] [ranks]"""
import glob, hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fulgor_amd import synth
from fulgor_amd.reads import ReadGenerator
from oracle.pyoracle import parse_compressed
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
b, o = ReadGenerator(g, raw_sequences=extra).generate(0, n, 150, 21)
path = "/dev/shm/ranks_%d.fq" % os.getpid()
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
del rec
outs = []
try:
    for fmt, m in (("compressed", n), ("binary", min(n, 300_000))):
        q = path
        if m < n:  # (3 KB per read: a smaller file for the verbose formats)
            q = path + ".small"
            with open(path, "rb") as f, open(q, "wb") as gq:
                gq.write(f.read(m * 316))
            outs.append(q)
        res = {}
        for world in (1, ranks):
            out = "/dev/shm/ranks_out_%d_%s_%d" % (os.getpid(), fmt, world)
            outs.append(out)
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "fulgor_amd", "pseudoalign", "-i", fg, "-q", q, "-o", out, "--format", fmt, "--gpus", str(world)],
                               cwd=ROOT, env=dict(os.environ, FULGOR_SHARE_GPU="1"), capture_output=True, text=True, timeout=1200)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr[-2000:]
            data = open(out, "rb").read()
            res[world] = data
            print("%-10s %d reads, %d rank(s): %.1f s for the command (start, index load, query), %d bytes, sha256 %s" % (fmt, m, world, dt, len(data), hashlib.sha256(data).hexdigest()[:16]), flush=True)
        if fmt == "compressed":
            a, c = parse_compressed(res[1]), parse_compressed(res[ranks])
            same = all(np.array_equal(x, y) for x, y in zip(a, c))
        else:
            same = res[1] == res[ranks]
        print("%-10s %d ranks against one: %s" % (fmt, ranks, "SAME" if same else "DIFFERENT"), flush=True)
        if not same:
            raise SystemExit(1)
finally:
    for p in [path] + outs:
        if os.path.exists(p):
            os.remove(p)
