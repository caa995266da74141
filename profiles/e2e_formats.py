"""The streamed command-line path with ascii / binary records (the reference's default is ascii: 3 KB of text per read on the bench
workload, 17 times the compressed records): reads/s and output GB/s by batch size. python profiles/e2e_formats.py [n reads]"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import FastxReader, ReadGenerator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
b, o = ReadGenerator(g, raw_sequences=extra).generate(0, n, 150, 42)
path = "/dev/shm/e2e_fmt_%d.fq" % os.getpid()
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
try:
    ix = fulgor_amd.Index(fg, device=0)
    for fmt, name in ((0, "ascii"), (1, "binary"), (2, "compressed")):
        for batch in (1 << 14, 1 << 15, 1 << 16, 1 << 17, 1 << 18, 0):
            for workers in (5, 3):
                ts = []
                for rep in range(3):
                    t0 = time.perf_counter()
                    rd = FastxReader(path, copy=False, threads=24)
                    fd = os.open("/dev/null", os.O_WRONLY)
                    got, mapped = ix.pseudoalign_stream(rd, fd, 0, 0.0, fmt, 0, True, batch, workers)
                    os.close(fd)
                    rd.close()
                    ts.append(time.perf_counter() - t0)
                rep_ = ix.last_stream_report().splitlines()[0]
                out_bytes = int(rep_.split(" ms, ")[1].split()[0])
                print("%-10s batch %7d workers %d: first %.0f ms, best %.1f ms = %.1f M reads/s, %.1f GB/s of records (%.0f bytes per read)"
                      % (name, batch, workers, ts[0] * 1e3, min(ts) * 1e3, n / min(ts) / 1e6, out_bytes / min(ts) / 1e9, out_bytes / n), flush=True)
finally:
    os.remove(path)
