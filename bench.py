#!/usr/bin/env python3
"""bench.py — pseudoaligned reads/s of the MI355X engine (BASELINE.json metric), one process per GPU.

A step = one pass of the hot path (k-mer lookup -> colour-set ids -> full-intersection / threshold-union
-> CSR colour lists -> per-colour hit counts) over this rank's reads, which are resident in HBM when the
timed region starts; the per-colour hit counts are all-reduced over RCCL at the end of every step.
Weak scaling: every rank gets `--reads` reads (rank r owns global reads [r*reads, (r+1)*reads)); the
index is replicated. Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def s10_genomes():
    g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
    assert len(g) == 10, "tests/data/salmonella_10 is incomplete"
    return g


def prepare_workload(name, rank):
    """returns (index path, ReadGenerator, description). Rank 0 builds missing caches; others wait."""
    import __graft_entry__ as ge
    from fulgor_amd.reads import ReadGenerator
    if name == "s10":
        fg, _ = ge._s10_index()
        return fg, ReadGenerator(s10_genomes()), "salmonella_10 (real genomes, 10 colours, 6.9M 31-mers, 171 colour sets)"
    if name == "s4546syn":
        from fulgor_amd import synth
        fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), s10_genomes())
        return fg, ReadGenerator(s10_genomes(), raw_sequences=extra), synth.DESCRIPTION
    raise SystemExit("unknown workload %s" % name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("FULGOR_BENCH_WORKLOAD", "s4546syn"), choices=["s4546syn", "s10"])
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the BASELINE config size)")
    ap.add_argument("--algo", default="full-intersection", choices=["full-intersection", "threshold-union"])
    ap.add_argument("--tau", type=float, default=0.8)
    ap.add_argument("--chunk", type=int, default=10_000_000,
                    help="reads per kernel pass (default: the whole read set of the BASELINE config in one pass; the result buffers of a "
                         "10 M-read pass take 40 GB (full intersection) to 95 GB (threshold union) of the 288 GB)")
    ap.add_argument("--index-type", default="hybrid", choices=["hybrid", "diff", "meta", "meta-diff"],
                    help="colour-set codec (fur / dfur / mfur / mdfur of the reference)")
    ap.add_argument("--partition-size", type=int, default=160)
    ap.add_argument("--cluster-size", type=int, default=16)
    ap.add_argument("--streams", type=int, default=1,
                    help="passes in flight per GPU: chunk i runs on stream i %% streams (own result buffers, own host thread)")
    ap.add_argument("--read-len", type=int, default=150, help="read length in bases (the metric is quoted on 150)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--order", type=int, default=1, help="0: take the reads of a pass in file order (fgpu_tune; A/B measurements)")
    ap.add_argument("--small", type=int, default=1, help="0: a bitmap row for every result (fgpu_tune; A/B measurements)")
    ap.add_argument("--rows", type=int, default=1, help="0: full intersection on the packed blocks, not on dense rows (fgpu_tune; A/B measurements)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import __graft_entry__ as ge
    # Rank 0 builds the library and the index caches BEFORE any rank joins the process group; the other ranks wait for a marker
    # file, not inside a collective: a slow build (cold box, 40 s to minutes for the synthetic index) cannot run into the
    # process-group timeout.
    marker = os.path.join(ROOT, "data", ".bench_ready_%s_%s" % (args.workload, os.environ.get("MASTER_PORT", "0")))
    if rank == 0:
        if os.path.exists(marker):
            os.remove(marker)
        ge.build()
        fg, gen, desc = prepare_workload(args.workload, rank)
        os.makedirs(os.path.dirname(marker), exist_ok=True)
        open(marker, "w").close()
    else:
        t_wait = time.time()
        while not os.path.exists(marker):
            if time.time() - t_wait > 3600:
                raise SystemExit("rank 0 did not finish preparing the workload within an hour")
            time.sleep(0.5)
        fg, gen, desc = prepare_workload(args.workload, rank)
    import fulgor_amd  # binds libfulgor_gpu.so to torch's HIP runtime (fulgor_amd/_native.py)
    import torch
    import torch.distributed as dist
    # FULGOR_BENCH_SHARE_GPU=1 (test only): all ranks use cuda:0 and gloo, to exercise the multi-rank control
    # flow on a single-GPU box; the real path is one GPU per rank over RCCL ("nccl" backend on ROCm)
    share = os.environ.get("FULGOR_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        if share:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=30))
        dist.barrier()
        if rank == 0 and os.path.exists(marker):
            os.remove(marker)

    default_reads = {"s10": 1_000_000, "s4546syn": 10_000_000}
    n_reads = args.reads or default_reads[args.workload]

    ix = fulgor_amd.Index(fg, device=local_rank)
    if not args.order or not args.small or not args.rows:
        ix.tune(order_min_reads=None if args.order else -1, small_results=bool(args.small), dense_rows=bool(args.rows))
    itype = {"hybrid": 0, "diff": 1, "meta": 2, "meta-diff": 3}[args.index_type]
    if itype:
        ix.convert(itype, args.partition_size, args.cluster_size)
        desc += "; colour sets re-encoded as %s (partitions of %d colours, clusters of %d sets)" % (
            args.index_type, args.partition_size, args.cluster_size)
    algo = fulgor_amd.FULL_INTERSECTION if args.algo == "full-intersection" else fulgor_amd.THRESHOLD_UNION
    bases, offs = gen.generate(rank * n_reads, n_reads, args.read_len, 42)
    reads = ix.upload_reads(bases, offs)
    results = [ix.new_result() for _ in range(max(1, args.streams))]
    res = results[0]
    ncol = ix.num_colors()
    hits = torch.zeros(ncol + 2, dtype=torch.int64, device="cuda:%d" % local_rank)
    chunks = [(first, min(args.chunk, n_reads - first)) for first in range(0, n_reads, args.chunk)]

    def worker(w):
        for i in range(w, len(chunks), len(results)):
            ix.run(reads, results[w], algo, args.tau, chunks[i][0], chunks[i][1])
            results[w].accumulate_hits(hits.data_ptr())

    def step():
        hits.zero_()
        torch.cuda.synchronize()
        if len(results) == 1:
            worker(0)
        else:  # the C ABI calls release the GIL; every result owns its HIP stream
            import threading
            ts = [threading.Thread(target=worker, args=(w,)) for w in range(len(results))]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        if world > 1:  # RCCL: per-colour hit counts + {reads, mapped}
            if share:
                h_cpu = hits.cpu()
                dist.all_reduce(h_cpu)
                hits.copy_(h_cpu)
            else:
                dist.all_reduce(hits)

    for _ in range(args.warmup):
        step()
    ix.timing_enable(True)
    ix.timing_reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda:%d" % local_rank)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    timing = ix.timing()
    ix.timing_enable(False)
    h = hits.cpu().numpy()
    total_reads_job, mapped_job = int(h[ncol]), int(h[ncol + 1])

    # algorithmic bytes of one step on this rank, from the resident per-read id lists / sizes
    acct = {"lists": 0, "output": 0, "lookup": 0}
    total_colors = 0
    for first in range(0, n_reads, args.chunk):
        cnt = min(args.chunk, n_reads - first)
        ix.run(reads, res, algo, args.tau, first, cnt)
        a = res.algorithmic_bytes()
        for k_ in acct:
            acct[k_] += a[k_]
        total_colors += res.sizes()[1]

    if rank == 0:
        stage_kernel = "k2a_intersect" if algo == fulgor_amd.FULL_INTERSECTION else "k3a_union"
        kbytes = {"k1_lookup": acct["lookup"], stage_kernel: acct["lists"], "k2b_expand": acct["output"]}
        cand = {k_: timing[k_] for k_ in kbytes if timing[k_][1] > 0}
        dom = max(cand, key=lambda k_: cand[k_][0])
        dom_ms, dom_launches = cand[dom]
        launches_per_step = dom_launches / args.steps
        avg_ms = dom_ms / dom_launches
        achieved = kbytes[dom] / launches_per_step / (avg_ms * 1e-3) / 1e9
        kernels = {k_: {"avg_ms": round(v[0] / v[1], 4), "launches": v[1]} for k_, v in timing.items() if v[1]}
        stage_ms = sum(timing[k_][0] for k_ in (stage_kernel, "k_order", "scan", "k2b_expand")) / args.steps
        # HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
        # build (profiles/traffic.json, made by profiles/traffic_from_pmc.py); only quoted for the configuration the
        # counters were collected on (same workload, codec and reads per launch), otherwise null.
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if (args.workload == "s4546syn" and itype == 0 and tj["reads_per_launch"] == args.chunk and n_reads >= args.chunk
                    and tj.get("algo", "full-intersection") == args.algo
                    and dom in tj["kernels"]):
                traffic, traffic_src = tj["kernels"][dom]["total"], tj["source"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "pseudoaligned reads/sec (%d bp, k=31)" % args.read_len,
            "value": round(world * n_reads * args.steps / elapsed, 1),
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%s, %s, %d synthetic %d bp reads per GPU (seed 42), k=31, chunk %d reads/pass"
                                   % (desc, args.algo + (" tau=%g" % args.tau if algo else ""), n_reads, args.read_len, args.chunk),
                       "index_replicated": True, "reads_per_gpu": n_reads, "streams": len(results),
                       "mapped_fraction": round(mapped_job / max(1, total_reads_job), 4),
                       "avg_colours_per_read": round(total_colors / n_reads, 2)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(kbytes[dom] / launches_per_step),
                         "avg_launch_ms": round(avg_ms, 4),
                         "stage": {"kernels": [stage_kernel, "k_order", "scan", "k2b_expand"],
                                   "bytes_per_step": acct["lists"] + acct["output"], "ms_per_step": round(stage_ms, 4),
                                   "achieved": round((acct["lists"] + acct["output"]) / (stage_ms * 1e-3) / 1e9, 2)}},
            "kernels": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["end_to_end"] = end_to_end(ix, bases, offs, algo, args.tau, min(n_reads, 1 << 20), 0)
            out["end_to_end_compressed"] = end_to_end(ix, bases, offs, algo, args.tau, min(n_reads, 1 << 20), 2)
            try:
                out["cli_end_to_end"] = cli_end_to_end(ix, bases, offs, algo, args.tau, min(n_reads, 10_000_000), args.read_len)
            except Exception as e:  # (no room for the FASTQ file, ...): the bench line must not depend on this leg
                out["cli_end_to_end"] = {"value": None, "error": str(e)[:200]}
            out["cpu_baseline"] = cpu_baseline(ix, bases, offs, algo, args.tau, itype, args.partition_size, args.cluster_size)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def end_to_end(ix, bases, offs, algo, tau, n, fmt):
    """PCIe-inclusive rate of one bounded pass (SURVEY §8d timing protocol): host-resident reads -> H2D -> kernels ->
    records formatted on the device (0 ascii, 2 the reference's compressed format) -> D2H of the output into host
    memory. Reported beside `value`, never as it."""
    import fulgor_amd
    b, o = bases[:int(offs[n])], offs[:n + 1]
    res = ix.new_result()
    best = None
    for _ in range(2):  # the second pass runs with warm buffers
        t0 = time.perf_counter()
        rd = ix.upload_reads(b, o)
        ix.run(rd, res, algo, tau)
        text = res.format_view(fmt, 0)
        dt = time.perf_counter() - t0
        rd.close()
        best = dt if best is None else min(best, dt)
    out_bytes = len(text)
    res.close()
    return {"value": round(n / best, 1), "unit": "reads/s", "reads": int(n), "output_bytes": int(out_bytes),
            "includes": "H2D of the reads, all kernels, device-side %s formatting, D2H of the output into a pinned, recycled "
                        "host buffer" % ("ascii" if fmt == 0 else "compressed")}


def cli_end_to_end(ix, bases, offs, algo, tau, n, read_len):
    """Wall clock of the command-line path (`python -m fulgor_amd pseudoalign`: driver.pseudoalign_sharded) on a bounded
    sample: an uncompressed FASTQ file on tmpfs -> parallel parse into pinned batches -> H2D -> kernels -> records in the
    reference's compressed format built on the device -> D2H -> /dev/null, three passes in flight. What the reference's own
    published figure measures (tools/pseudoalign.cpp:76-88), reported beside `value`, never as it. The index is open already."""
    import tempfile
    from fulgor_amd import driver
    import shutil
    need = n * (12 + 2 * read_len + 4) + (64 << 20)
    d = next((x for x in ("/dev/shm", tempfile.gettempdir(), os.path.join(ROOT, "data"))
              if os.path.isdir(x) and shutil.disk_usage(x).free > need), None)
    if d is None:
        raise RuntimeError("no directory with %d MB free for the FASTQ sample" % (need >> 20))
    path = os.path.join(d, "fulgor_bench_%d.fq" % os.getpid())
    try:
        width = 1 + 1 + 9 + 1  # "@r%09d\n"
        rec = np.empty((n, width + read_len + 3 + read_len + 1), dtype=np.uint8)
        ids = np.arange(n, dtype=np.int64)
        rec[:, 0], rec[:, 1], rec[:, width - 1] = ord("@"), ord("r"), ord("\n")
        for dgt in range(9):
            rec[:, 2 + dgt] = ord("0") + (ids // 10 ** (8 - dgt)) % 10
        rec[:, width:width + read_len] = np.asarray(bases[:n * read_len]).reshape(n, read_len)
        rec[:, width + read_len:width + read_len + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, width + read_len + 3:-1] = ord("I")
        rec[:, -1] = ord("\n")
        rec.tofile(path)
        size = os.path.getsize(path)
        del rec
        best = None
        for _ in range(2):  # the second run finds warm buffers
            t0 = time.perf_counter()
            got, mapped = driver.pseudoalign_sharded(lambda: ix, path, "/dev/null", algo, tau, "compressed")
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert got == n
    finally:
        if os.path.exists(path):
            os.remove(path)
    return {"value": round(n / best, 1), "unit": "reads/s", "reads": int(n), "fastq_bytes": int(size),
            "includes": "FASTQ file on tmpfs -> parse (parallel, native) -> pinned batches -> H2D -> kernels -> compressed records "
                        "built on the device -> D2H -> /dev/null; index already resident"}


def cpu_baseline(ix, bases, offs, algo, tau, itype=0, psize=160, csize=16):
    """The oracle (CPU restatement of the reference, same worker-pool threading) timed on the host cores
    on a bounded prefix of the same reads; kind = "port" because the reference binary cannot be built."""
    from oracle.pyoracle import OracleIndex
    orc = OracleIndex.from_export(ix.export())
    if itype:
        orc.convert(itype, psize, csize)
    host = os.cpu_count() or 1
    probe = min(len(offs) - 1, 20000)
    # the port does not scale to every hardware thread of a big host (it allocates per read, as the reference does): probe a
    # few thread counts on 20000 reads each and time the sample at the best one
    rates = {}
    for t in sorted({host, max(1, host // 2), max(1, host // 4), max(1, host // 8), 8}, reverse=True):
        sec, _, _ = orc.time_pseudoalign(bases[:int(offs[probe])], offs[:probe + 1], algo, tau, t)
        rates[t] = probe / max(sec, 1e-9)
    cores = max(rates, key=rates.get)
    n = int(min(len(offs) - 1, max(probe, rates[cores] * 12.0)))
    sec, mapped, _ = orc.time_pseudoalign(bases[:int(offs[n])], offs[:n + 1], algo, tau, cores)
    return {"value": round(n / sec, 1), "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d reads of rank 0's read set, %d worker threads (the best of the probed thread counts on this "
                      "%d-thread host), output discarded (%.1f s)" % (n, cores, host, sec),
            "per_thread": round(n / sec / cores, 1),
            "thread_probe": {str(t): round(r, 1) for t, r in sorted(rates.items())},
            "context": "the reference's README quotes 50.6 k reads/s at -t 8 (about 7 k reads/s per worker) for its own binary on "
                       "real reads and unnamed hardware; this port is a restatement with an exact k-mer hash map in place of SSHash, "
                       "not the reference binary"}


if __name__ == "__main__":
    main()
