"""Stress of the streamed loop's copy path: the same FASTQ file (n reads of the bench workload) through fgpu_pseudoalign_stream a few hundred
times with 1..8 workers and 1..32 parser threads at random; for a given batch size the compressed records are the same bytes whatever
the workers and threads (batches are cut from the chunk sequence deterministically): one digest per batch size, every run compared.
python profiles/stream_stress.py [n reads] [runs]"""
import glob, hashlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import FastxReader, ReadGenerator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
b, o = ReadGenerator(g, raw_sequences=extra).generate(0, n, 150, 7)
path = "/dev/shm/stress_%d.fq" % os.getpid()
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
out_path = "/dev/shm/stress_out_%d.bin" % os.getpid()
rng = np.random.default_rng(99)
digests = {}
try:
    ix = fulgor_amd.Index(fg, device=0)
    t_all = time.perf_counter()
    for it in range(runs):
        batch = int(rng.choice([1 << 14, 1 << 16, 1 << 18]))
        workers, threads = int(rng.integers(1, 9)), int(rng.choice([1, 4, 16, 24, 32]))
        algo, tau = ((0, 0.0), (1, 0.8))[it % 5 == 4]
        rd = FastxReader(path, copy=False, threads=threads)
        fd = os.open(out_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
        got, mapped = ix.pseudoalign_stream(rd, fd, algo, tau, 2, 0, True, batch, workers)
        os.close(fd)
        rd.close()
        assert got == n
        h = hashlib.sha256(open(out_path, "rb").read()).hexdigest()
        key = (batch, algo)
        if digests.setdefault(key, h) != h:
            raise SystemExit("run %d (batch %d, %d workers, %d threads, algo %d): records differ from the first run with this batch size" % (it, batch, workers, threads, algo))
    print("%d runs of %d reads in %.1f s: every run gave the bytes of the first run with its batch size (%d settings of batch x algorithm)"
          % (runs, n, time.perf_counter() - t_all, len(digests)))
    print(ix.last_stream_report().splitlines()[2])
finally:
    for p in (path, out_path):
        if os.path.exists(p):
            os.remove(p)
