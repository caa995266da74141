"""A/B of settings of the streamed command-line path INSIDE one process (run-to-run and box-to-box spread is larger than most effects):
the variants take turns, `rounds` times; per variant the median and the minimum. python profiles/e2e_ab.py [n reads] [rounds]
A variant is name:ENV=VALUE,...:threads:workers:batch"""
import glob, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import FastxReader, ReadGenerator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
variants = sys.argv[3:] or ["sched:FULGOR_READER_AFFINITY=0:24:5:262144", "file:FULGOR_READER_AFFINITY=file:24:5:262144", "device:FULGOR_READER_AFFINITY=device:24:5:262144"]
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
b, o = ReadGenerator(g, raw_sequences=extra).generate(0, n, 150, 42)
path = "/dev/shm/e2e_ab_%d.fq" % os.getpid()
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
    times = {v: [] for v in variants}
    for r in range(rounds + 1):
        for v in variants:
            name, envs, threads, workers, batch = v.split(":")
            saved = {}
            for kv in envs.split(","):
                if kv:
                    k_, _, val = kv.partition("=")
                    saved[k_] = os.environ.get(k_)
                    os.environ[k_] = val
            t0 = time.perf_counter()
            rd = FastxReader(path, copy=False, threads=int(threads))
            fd = os.open("/dev/null", os.O_WRONLY)
            got, mapped = ix.pseudoalign_stream(rd, fd, 0, 0.0, 2, 0, True, int(batch), int(workers))
            os.close(fd)
            rd.close()
            dt = time.perf_counter() - t0
            for k_, val in saved.items():
                if val is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = val
            if r:  # (round 0 warms the buffers)
                times[v].append(dt)
    for v in variants:
        t = sorted(times[v])
        print("%-60s median %.1f ms (%.0f M reads/s)  min %.1f  max %.1f  all: %s" % (v, statistics.median(t) * 1e3, n / statistics.median(t) / 1e6, t[0] * 1e3, t[-1] * 1e3, " ".join("%.0f" % (x * 1e3) for x in times[v])))
finally:
    os.remove(path)
