"""Which of the secondary workloads of bench.py leaves the streamed command-line path slower behind it (190 -> 130 M reads/s in the bench process)?
One process: the streamed run 4x, then one secondary-like action at a time, the streamed run 3x after each. python profiles/e2e_after_what.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, fulgor_amd
from fulgor_amd import driver
import torch
n = 10_000_000
w = bench.Workload("s4546syn", 0, 0, n, 150, "hybrid", 160, 16)
path = "/dev/shm/e2e_after_%d.fq" % os.getpid()
rec = np.empty((n, 316), dtype=np.uint8)
ids = np.arange(n, dtype=np.int64)
rec[:, 0], rec[:, 1], rec[:, 11] = ord("@"), ord("r"), ord("\n")
for d in range(9):
    rec[:, 2 + d] = ord("0") + (ids // 10 ** (8 - d)) % 10
rec[:, 12:162] = np.asarray(w.bases[:n * 150]).reshape(n, 150)
rec[:, 162:165] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 165:-1] = ord("I")
rec[:, -1] = ord("\n")
rec.tofile(path)
del rec


import glob, statistics, threading
_cards = [c for c in glob.glob("/sys/class/drm/card*/device") if os.path.exists(c + "/mem_info_vram_used")]
_card = max(_cards, key=lambda c: int(open(c + "/mem_info_vram_used").read() or 0)) if _cards else None
_hw = (glob.glob(_card + "/hwmon/hwmon*") or [None])[0] if _card else None


def _cur(name):
    try:
        t = open(_card + "/" + name).read()
    except OSError:
        return ""
    cur = [l for l in t.splitlines() if l.endswith("*")]
    return cur[0].split(":")[1].strip(" *") if cur else ""


def stream(tag, k=3):
    ts = []
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            samples.append({n_: _cur(n_) for n_ in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "pp_dpm_pcie")})
            time.sleep(0.005)
    th = threading.Thread(target=sampler)
    if _card:
        th.start()
    for _ in range(k):
        t0 = time.perf_counter()
        driver.pseudoalign_sharded(lambda: w.ix, path, "/dev/null", 0, 0.0, "compressed")
        ts.append((time.perf_counter() - t0) * 1e3)
        rep = w.ix.last_stream_report().splitlines()[1]
    stop[0] = True
    if _card:
        th.join()
    free, total = torch.cuda.mem_get_info(0)
    clocks = []
    for n_ in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "pp_dpm_pcie"):
        vals = [x[n_] for x in samples if x.get(n_)]
        if vals:
            clocks.append("%s %s" % (n_[7:], max(set(vals), key=vals.count)))
    print("%-60s %s ms | %s | %s" % (tag, " ".join("%.0f" % t for t in ts), rep[rep.index("host buffers") + 41:], "; ".join(clocks)), flush=True)
    if os.environ.get("E2E_AFTER_STAGES"):
        w.ix.timing_enable(True)
        w.ix.timing_reset()
        t0 = time.perf_counter()
        driver.pseudoalign_sharded(lambda: w.ix, path, "/dev/null", 0, 0.0, "compressed")
        dt = (time.perf_counter() - t0) * 1e3
        tm = w.ix.timing()
        w.ix.timing_enable(False)
        nb = max(1, tm["k1_lookup"][1])
        print("    with event timing: %.0f ms; per batch of 2^18 (ms): %s" % (dt, ", ".join("%s %.3f" % (k_, v[0] / nb) for k_, v in tm.items() if v[1])), flush=True)
        rep = w.ix.last_stream_report().splitlines()
        for l in rep[3:3 + 14:1]:
            print("      " + l)


class A:
    chunk, rows, partition_size, cluster_size, read_len = 10_000_000, None, 160, 16, 150


try:
    if os.environ.get("E2E_AFTER_FINE"):
        stream("fresh", 4)
        r = w.ix.new_result(); r.close()
        stream("after a result created and closed")
        hits = torch.zeros(4548, dtype=torch.int64, device="cuda:0"); torch.cuda.synchronize()
        stream("after a torch tensor on the device")
        r = w.ix.new_result(); w.ix.run(w.reads, r, 0, 0.0, 0, 1_000_000); r.close()
        stream("after one pass of 1 M reads (no expansion)")
        r = w.ix.new_result(); w.ix.run(w.reads, r, 0, 0.0, 0, 10_000_000); r.close()
        stream("after one pass of 10 M reads (no expansion)")
        r = w.ix.new_result(); w.ix.run(w.reads, r, 0, 0.0, 0, 10_000_000); r.expand(); r.close()
        stream("after one pass of 10 M reads + expansion (24 GB of colours)")
        r = w.ix.new_result(); w.ix.run(w.reads, r, 0, 0.0, 0, 10_000_000); r.expand(); r.accumulate_hits(hits.data_ptr()); r.close()
        stream("after pass + expansion + hit vector")
        w.ix.timing_enable(True); r = w.ix.new_result(); w.ix.run(w.reads, r, 0, 0.0, 0, 10_000_000); r.close(); w.ix.timing_enable(False)
        stream("after a pass with HIP-event timing on")
        raise SystemExit(0)
    stream("fresh", 4)
    bench.measure(w, 0, 0.0, A.chunk, 3, 1, 1, 0)
    stream("after 4 full-intersection steps (10 M reads, expand)")
    bench.measure(w, 1, 0.8, A.chunk, 3, 1, 1, 0)
    stream("after 4 threshold-union steps (95 GB of result buffers)")
    w.ix.tune(dense_rows=False)
    full = w.n_reads
    w.n_reads = 5_000_000
    bench.measure(w, 0, 0.0, A.chunk, 3, 1, 1, 0)
    w.ix.tune(dense_rows=True)
    stream("after the packed-block kernels on 5 M reads")
    w.ix.convert(3, 160, 16)
    bench.measure(w, 0, 0.0, A.chunk, 3, 1, 1, 0)
    w.ix.convert(0)
    w.n_reads = full
    stream("after convert(meta-diff) + steps + convert(hybrid)")
    wc = bench.Workload("s4546core", 0, 0, 5_000_000, 150, "hybrid", 0, 0)
    bench.measure(wc, 0, 0.0, A.chunk, 3, 1, 1, 0)
    wc.close()
    stream("after a second index (core-heavy) opened, run, closed")
    torch.cuda.empty_cache()
    stream("after torch.cuda.empty_cache()")
finally:
    os.remove(path)
