"""The command-line path stage by stage (GPU box): a FASTQ file of n reads on tmpfs -> reader alone (record count = grammar walk
without copies; chunk stream = parse into pinned chunks) at several thread counts -> the native worker loop
(fgpu_pseudoalign_stream, compressed records to /dev/null) over a grid of parser threads / workers / range sizes / batch sizes,
with the timeline of the last run and the per-kernel / per-copy HIP-event times.
usage: python profiles/e2e_stream.py [n reads] [quick]"""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import FastxReader, ReadGenerator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
quick = len(sys.argv) > 2
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
b, o = ReadGenerator(g, raw_sequences=extra).generate(0, n, 150, 42)
path = "/dev/shm/e2e_%d.fq" % os.getpid()
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
del rec, b, o
size = os.path.getsize(path)
print("file: %d reads, %.2f GB of text on tmpfs; host: %d hardware threads" % (n, size / 1e9, os.cpu_count()))


def run(ix, threads, workers, batch, fmt=2, out="/dev/null"):
    t0 = time.perf_counter()
    rd = FastxReader(path, copy=False, threads=threads)
    fd = os.open(out, os.O_WRONLY)
    got, mapped = ix.pseudoalign_stream(rd, fd, 0, 0.0, fmt, 0, True, batch, workers)
    os.close(fd)
    rd.close()
    dt = time.perf_counter() - t0
    assert got == n
    return dt


try:
    ix = fulgor_amd.Index(fg, device=0)  # (installs the pinned allocator: the reader's chunks are pinned from here on)
    for threads in ((8, 32, 64) if quick else (1, 8, 16, 32, 64, 96, 128)):
        t0 = time.perf_counter()
        rd = FastxReader(path, copy=False, threads=threads)
        t1 = time.perf_counter()
        c = rd.count()
        t2 = time.perf_counter()
        rd.close()
        t3 = time.perf_counter()
        assert c == n
        print("count (grammar walk, no copies), %3d threads: open %.1f ms, count %.1f ms = %.1f M reads/s = %.1f GB/s of text, close %.1f ms"
              % (threads, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n / (t2 - t1) / 1e6, size / (t2 - t1) / 1e9, (t3 - t2) * 1e3))
    for threads in ((8, 32, 64) if quick else (1, 8, 16, 32, 64, 96, 128)):
        for rep in range(2):
            t0 = time.perf_counter()
            rd = FastxReader(path, copy=False, threads=threads)
            fd = os.open("/dev/null", os.O_WRONLY)
            # the worker loop with nothing to do on the device is not available: the parse rate is read off a full run's parser statistics below;
            # here: batches through the copying interface (fgpu_fastx_next), as the k-mer tools and --deduplicate use it
            tot = sum(len(of) - 1 for _, of in rd)
            os.close(fd)
            rd.close()
            dt = time.perf_counter() - t0
        print("fgpu_fastx_next (parse + gather into pinned batches), %3d threads: %.1f ms  %.1f M reads/s  %.1f GB/s of text" % (threads, dt * 1e3, tot / dt / 1e6, size / dt / 1e9))
    grid = [(32, 4, 1 << 19), (24, 4, 1 << 19)] if quick else [(t, w, b_) for b_ in (1 << 17, 1 << 18, 1 << 19) for w in (3, 4, 5, 6) for t in (16, 24, 32, 48)]
    best = None
    run(ix, 32, 6, 1 << 19)  # (pins the host buffers, sizes the device buffers)
    for threads, workers, batch in grid:
        ts = [run(ix, threads, workers, batch) for _ in range(4)]
        dt = min(ts)
        print("stream compressed: %3d parser threads, %d workers, batch %7d: %.1f ms  %.1f M reads/s (runs: %s)"
              % (threads, workers, batch, dt * 1e3, n / dt / 1e6, " ".join("%.1f" % (t * 1e3) for t in ts)))
        if best is None or dt < best[0]:
            best = (dt, threads, workers, batch)
    _, threads, workers, batch = best
    for kb in (2048, 4096, 16384):
        os.environ["FULGOR_READER_RANGE_KB"] = str(kb)
        ts = [run(ix, threads, workers, batch) for _ in range(4)]
        print("stream compressed, ranges of %5d KB (%d threads, %d workers, batch %d): %.1f ms  %.1f M reads/s" % (kb, threads, workers, batch, min(ts) * 1e3, n / min(ts) / 1e6))
    del os.environ["FULGOR_READER_RANGE_KB"]
    ix.timing_enable(True)
    ix.timing_reset()
    dt = run(ix, threads, workers, batch)
    tm = ix.timing()
    ix.timing_enable(False)
    print("\nbest setting with HIP-event timing on (%d threads, %d workers, batch %d): %.1f ms  %.1f M reads/s" % (threads, workers, batch, dt * 1e3, n / dt / 1e6))
    nb = max(1, tm["k1_lookup"][1])
    print("per batch (avg over %d batches), ms: " % nb + ", ".join("%s %.3f" % (k_, v[0] / nb) for k_, v in tm.items() if v[1]))
    print(ix.last_stream_report())
    for fmt, name in ((1, "binary"), (0, "ascii")):
        m = min(n, 2_000_000)
        print("(%s output of the whole file: %.1f M reads/s)" % (name, n / run(ix, threads, workers, batch, fmt) / 1e6))
finally:
    os.remove(path)
