"""Where the command-line path spends its time (GPU box): reader alone, then the pipelined worker loop, on a FASTQ file
of n reads on tmpfs. usage: python profiles/ingest_breakdown.py [n reads] [batch]"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fulgor_amd
from fulgor_amd import driver, synth
from fulgor_amd.reads import FastxReader, ReadGenerator
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
b, o = ReadGenerator(g, raw_sequences=extra).generate(0, n, 150, 42)
path = "/dev/shm/ingest_%d.fq" % os.getpid()
rec = np.empty((n, 12 + 150 + 3 + 150 + 1), dtype=np.uint8)
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
try:
    for threads in (1, 8, 32, 64):
        t0 = time.perf_counter()
        rd = FastxReader(path, batch=batch, copy=False, threads=threads)
        tot = sum(len(of) - 1 for _, of in rd)
        rd.close()
        dt = time.perf_counter() - t0
        print("reader only, %2d threads: %.3f s  %.1f M reads/s  %.2f GB/s of text" % (threads, dt, tot / dt / 1e6, os.path.getsize(path) / dt / 1e9))
    ix = fulgor_amd.Index(fg, device=0)
    for fmt in ("compressed", "binary"):
        for rep in range(2):
            t0 = time.perf_counter()
            rd = FastxReader(path, batch=batch, copy=False, threads=32)
            with open("/dev/null", "wb") as out:
                got, mapped = driver.pseudoalign_stream(ix, rd, sink=out, fmt=fmt)
            rd.close()
            dt = time.perf_counter() - t0
            print("pipeline %-10s run %d: %.3f s  %.1f M reads/s" % (fmt, rep, dt, got / dt / 1e6))
    # stages of one batch, serial
    rd = FastxReader(path, batch=batch, copy=False, threads=32)
    bases, offs = next(iter(rd))
    res = ix.new_result()
    for rep in range(3):
        t0 = time.perf_counter(); reads = ix.upload_reads(bases, offs); t1 = time.perf_counter()
        ix.run(reads, res, 0, 0.0); t2 = time.perf_counter()
        v = res.format_view(2, 0); t3 = time.perf_counter()
        reads.close(); t4 = time.perf_counter()
        print("one batch of %d reads: upload %.1f ms, run %.1f ms, format+D2H %.1f ms (%d bytes), free %.1f ms" % (len(offs) - 1, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(v), (t4 - t3) * 1e3))
finally:
    os.remove(path)
