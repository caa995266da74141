"""one streamed run of the command-line path on an n-read FASTQ file on tmpfs (for rocprofv3): python profiles/e2e_once.py [n] [threads] [workers] [batch] [runs]"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import FastxReader, ReadGenerator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
threads, workers, batch, runs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 32), (3, 4), (4, 1 << 19), (5, 3)))
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
b, o = ReadGenerator(g, raw_sequences=extra).generate(0, n, 150, 42)
path = "/dev/shm/e2e_once_%d.fq" % os.getpid()
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
if os.environ.get("E2E_TORCH"):  # (what bench.py has in its process: torch's HIP context, a tensor, an all-reduce-sized copy)
    import torch
    t = torch.zeros(4548, dtype=torch.int64, device="cuda:0")
    torch.cuda.synchronize()
    if os.environ["E2E_TORCH"] == "2":
        big = torch.empty(3 << 30, dtype=torch.uint8, device="cuda:0")
        host = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
try:
    ix = fulgor_amd.Index(fg, device=0)
    reps = []
    if os.environ.get("E2E_BIND"):  # the calling thread (and the threads it starts from here on) onto the first hardware thread of every core of one node
        node = int(os.environ["E2E_BIND"])
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b_ = part.partition("-")
            cpus.update(range(int(a), int(b_ or a) + 1))
        keep = {c for c in cpus if min(int(x) for x in open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().replace("-", ",").split(",")) == c}
        os.sched_setaffinity(0, keep & os.sched_getaffinity(0))
        print("bound to %d cpus of node %d" % (len(os.sched_getaffinity(0)), node))
    for r in range(runs):
        time.sleep(float(os.environ.get("E2E_SLEEP", "0")))
        t0 = time.perf_counter()
        rd = FastxReader(path, copy=False, threads=threads)
        fd = os.open("/dev/null", os.O_WRONLY)
        got, mapped = ix.pseudoalign_stream(rd, fd, 0, 0.0, 2, 0, True, batch, workers)
        os.close(fd)
        rd.close()
        dt = time.perf_counter() - t0
        print("run %d: %d reads in %.1f ms = %.1f M reads/s" % (r, got, dt * 1e3, got / dt / 1e6))
        reps.append((dt, r, ix.last_stream_report()))
    for dt_, r_, rep_ in reps:
        print("run %d: %s" % (r_, rep_.splitlines()[1][rep_.splitlines()[1].index("per thread"):]))
    if os.environ.get("E2E_COLD"):
        print("first run of the process (%d):\n%s" % (reps[0][1], reps[0][2]))
    steady = sorted(reps[2:] or reps)
    print("fastest steady run (%d):\n%s" % (steady[0][1], steady[0][2]))
    if len(steady) > 1:
        print("slowest steady run (%d):\n%s" % (steady[-1][1], steady[-1][2]))
finally:
    os.remove(path)
