"""Where the pipelined worker loop of the command-line path waits (GPU box): time of the main thread in the reader, time blocked
on the oldest pass, and the stages inside the passes. usage: python profiles/ingest_overlap.py [n reads] [batch] [inflight] [threads]"""
import glob, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fulgor_amd
from fulgor_amd import driver, synth
from fulgor_amd.reads import FastxReader, ReadGenerator
from collections import deque
from concurrent.futures import ThreadPoolExecutor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 19
inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 3
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
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
    ix = fulgor_amd.Index(fg, device=0)
    for rep in range(3):
        results = [ix.new_result() for _ in range(inflight)]
        st = {"upload": 0.0, "run": 0.0, "format": 0.0, "close": 0.0}
        lock = threading.Lock()

        def one_pass(slot, bases, offs, id0):
            res = results[slot]
            t0 = time.perf_counter(); reads = ix.upload_reads(bases, offs)
            t1 = time.perf_counter(); ix.run(reads, res, 0, 0.0)
            t2 = time.perf_counter(); view = res.format_view(2, id0)
            t3 = time.perf_counter(); reads.close()
            t4 = time.perf_counter()
            with lock:
                st["upload"] += t1 - t0; st["run"] += t2 - t1; st["format"] += t3 - t2; st["close"] += t4 - t3
            return view

        t_next = t_wait = 0.0
        pending = deque()
        T0 = time.perf_counter()
        rd = FastxReader(path, batch=batch, copy=False, threads=threads)
        it = iter(rd)
        t_open = time.perf_counter() - T0
        got = 0
        with ThreadPoolExecutor(max_workers=inflight) as pool, open("/dev/null", "wb") as out:
            slot = 0
            while True:
                t0 = time.perf_counter()
                try:
                    bases, offs = next(it)
                except StopIteration:
                    t_next += time.perf_counter() - t0
                    break
                t_next += time.perf_counter() - t0
                if len(pending) == inflight:
                    t0 = time.perf_counter(); out.write(pending.popleft().result()); t_wait += time.perf_counter() - t0
                pending.append(pool.submit(one_pass, slot, bases, offs, got))
                slot = (slot + 1) % inflight
                got += len(offs) - 1
            t0 = time.perf_counter()
            while pending:
                out.write(pending.popleft().result())
            t_tail = time.perf_counter() - t0
        t0 = time.perf_counter()
        rd.close()
        t_close = time.perf_counter() - t0
        dt = time.perf_counter() - T0
        for r in results:
            r.close()
        print("run %d: %.3f s  %.1f M reads/s | open %.3f s, close %.3f s | main thread: reader %.3f s, blocked on the oldest pass %.3f s, tail %.3f s | passes (sum over %d): upload %.3f run %.3f format+D2H %.3f close %.3f" % (
            rep, dt, got / dt / 1e6, t_open, t_close, t_next, t_wait, t_tail, (got + batch - 1) // batch, st["upload"], st["run"], st["format"], st["close"]))
finally:
    os.remove(path)
