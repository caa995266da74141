"""`fulgor pseudoalign`-compatible command line (tools/pseudoalign.cpp:228-369): same flags, same exit
codes, same summary lines; the index argument is a dump basename or an .fgidx container."""
import argparse

import os
import sys
import time

import numpy as np

_T_IMPORT = time.time()

from . import driver  # noqa: E402
from .index import FULL_INTERSECTION, THRESHOLD_UNION, Index  # noqa: E402
from .reads import FastxReader  # noqa: E402


def _mark(what):
    """FULGOR_CLI_TIMELINE=<epoch seconds at which the command was started>: the stages of one command on stderr"""
    t0 = os.environ.get("FULGOR_CLI_TIMELINE")
    if t0:
        print("[cli] +%.3f s %s" % (time.time() - float(t0), what), file=sys.stderr, flush=True)


def _launch_ranks(argv, gpus):
    """--gpus N outside a launcher: start one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torchrun
    exports them) running this same command line, wait for all of them"""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-m", "fulgor_amd", "pseudoalign"] + list(argv), env=env))
    # the first rank that fails takes the others down: they would wait in a collective until the process-group timeout
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.1)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = 1
                for q in live:
                    q.terminate()
    if rc:  # no half-written output: the parts of the ranks that got that far
        out = None
        for i, a in enumerate(argv):
            if a == "-o" and i + 1 < len(argv):
                out = argv[i + 1]
        if out:
            for r in range(gpus):
                try:
                    os.remove("%s.part%d" % (out, r))
                except OSError:
                    pass
    return rc


def pseudoalign(argv):
    ap = argparse.ArgumentParser(prog="fulgor pseudoalign", add_help=True)
    ap.add_argument("-i", dest="index_filename", required=True, help="The Fulgor index (dump basename or .fgidx).")
    ap.add_argument("-q", dest="query_filename", required=True, help="Query filename in FASTA/FASTQ format (optionally gzipped).")
    ap.add_argument("-o", dest="output_filename", required=True, help="File where output will be written.")
    ap.add_argument("-t", dest="num_threads", type=int, default=0,
                    help="Threads that parse the query file (the reference's parser / worker threads); 0 = half of the host's.")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("-r", dest="threshold", type=float, default=None,
                    help="Threshold for threshold_union algorithm. It must be a float in (0.0,1.0].")
    ap.add_argument("--deduplicate", action="store_true")
    ap.add_argument("--format", dest="format", default="ascii")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1,
                    help="GPUs of this node to use: one process per GPU, the index replicated, every GPU takes one part of the query file.")
    try:
        a = ap.parse_args(argv)
    except SystemExit:
        return 1
    algo = FULL_INTERSECTION
    if a.threshold is not None:
        if a.threshold <= 0.0 or a.threshold > 1.0:
            print("threshold must be a float in (0.0,1.0]", file=sys.stderr)  # tools/pseudoalign.cpp:275-278
            return 1
        if a.deduplicate:
            print("Deduplication not available for threshold < 1.0. Remove --deduplicate flag.", file=sys.stderr)
            return 1
        algo = THRESHOLD_UNION
    if a.format not in driver.FORMATS:
        print("Unknown output format. Supported formats: ascii, binary, compressed.")  # tools/pseudoalign.cpp:317-320
        return 1
    if a.gpus < 1:
        print("--gpus must be positive", file=sys.stderr)
        return 1
    rank, world, local = driver.rank_env()
    if a.gpus > 1 and "RANK" not in os.environ:
        return _launch_ranks(argv, a.gpus)
    if world != a.gpus:
        print("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
        return 1
    if a.verbose and rank == 0:
        print(" ".join(["fulgor", "pseudoalign"] + list(argv)))
    if not os.path.exists(a.query_filename):
        print("cannot open " + a.query_filename, file=sys.stderr)
        return 1
    # FULGOR_SHARE_GPU=1 (test only): all ranks use --device and gloo, to exercise the multi-rank path on a one-GPU box
    share = os.environ.get("FULGOR_SHARE_GPU") == "1"
    device = a.device if (world == 1 or share) else local
    reduce_device = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if share or not torch.cuda.is_available():
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))  # RCCL on ROCm
            reduce_device = "cuda:%d" % device
    _mark("imports and arguments done (package import began at +%.3f s)" % (_T_IMPORT - float(os.environ.get("FULGOR_CLI_TIMELINE") or _T_IMPORT)))
    clock = {}

    def open_index():
        """essentials::load(index, index_filename) (tools/pseudoalign.cpp:338-341); the query clock starts behind it, where the
        reference starts its own (pseudoalign_orchestrator, tools/pseudoalign.cpp:59-60)"""
        if a.verbose and rank == 0:
            print("*** START: loading the index")
        t_open = time.time()
        ix = Index(a.index_filename, device=device)
        clock["open_s"] = time.time() - t_open
        if a.verbose and rank == 0:
            print("*** DONE: loading the index (%d millisec)" % (clock["open_s"] * 1000))
            print("performing queries from file '%s'..." % a.query_filename)
        _mark("index open (%.3f s)" % clock["open_s"])
        clock["t_query"] = time.time()
        clock["index"] = ix  # (alive until the process ends: closing it — gigabytes of device buffers, one by one — is the operating system's job at exit)
        return ix

    try:
        if a.deduplicate and os.environ.get("FULGOR_DEDUPLICATE_ON_HOST"):  # (the round-2 path: grouping in numpy; kept for A/B runs)
            index = open_index()
            batches = FastxReader(a.query_filename, batch=1 << 19, copy=False, threads=a.num_threads)
            with open(a.output_filename, "wb") as out:
                n, mapped = driver.pseudoalign_stream(index, batches, algo, 0.0, sink=out, fmt=a.format, deduplicate=True)
            batches.close()
        else:  # read id = 0-based file order (src/ps_utils.cpp:276,286); parsing overlaps with the GPU passes
            # (--deduplicate: the same streamed path with the device-side grouping switched on — every distinct list of colour-set ids
            # of a batch is intersected once; records leave in file order, whatever the format)
            n, mapped = driver.pseudoalign_sharded((lambda: open_index().tune(deduplicate=True)) if a.deduplicate else open_index, a.query_filename,
                                                   a.output_filename, algo, a.threshold or 0.0, a.format, rank, world,
                                                   io_threads=a.num_threads, device_for_reduce=reduce_device,
                                                   prepare_device=None if os.environ.get("FULGOR_NO_PREPARE") else device)
    except (RuntimeError, ValueError) as e:
        print(str(e), file=sys.stderr)
        return 1
    finally:
        if world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
    el = (time.time() - clock.get("t_query", time.time())) * 1000.0
    _mark("query done (%.3f s)" % (el / 1000))
    if os.environ.get("FULGOR_CLI_TIMELINE") and rank == 0 and not a.deduplicate:
        try:
            from . import _native
            import ctypes as C
            p = C.c_void_p()
            if _native.lib().fgpu_last_stream_report(C.byref(p)) == 0:
                print(C.string_at(p.value).decode(), file=sys.stderr)
                _native.lib().fgpu_free(p)
        except Exception:  # noqa: BLE001 — instrumentation only
            pass
    if a.verbose and rank == 0:  # tools/pseudoalign.cpp:79-88
        print("processed %d reads" % n)
        print("elapsed = %d millisec / %d sec / %d min / %g musec/read" % (el, el / 1000, el / 60000, el * 1000 / max(1, n)))
        print("num_mapped_reads %d/%d (%g%%)" % (mapped, n, mapped * 100.0 / max(1, n)))
    return 0


def _query_tool(argv, prog, tool):
    """common driver of the two per-k-mer tools (tools/kmer_conservation.cpp:58-127, tools/kmer_matches.cpp:57-126):
    -i index -q reads -o output [-t threads] [--verbose]; one output line per record, in file order. The lines are made by
    the native emitters (fgpu_kmer_emitter_*: lookup and counts on the device, text on the host's threads)."""
    ap = argparse.ArgumentParser(prog="fulgor " + prog, add_help=True)
    ap.add_argument("-i", dest="index_filename", required=True)
    ap.add_argument("-q", dest="query_filename", required=True)
    ap.add_argument("-o", dest="output_filename", required=True)
    ap.add_argument("-t", dest="num_threads", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    try:
        a = ap.parse_args(argv)
    except SystemExit:
        return 1
    if not os.path.exists(a.query_filename):
        print("error in opening the file '%s'" % a.query_filename, file=sys.stderr)
        return 1
    from .index import KmerEmitter
    try:
        index = Index(a.index_filename, device=a.device)
        # kmer-matches: one count per colour and record (the dense table of a batch stays below a quarter of a gigabyte)
        batch = 1 << 16 if tool == 0 else max(256, min(1 << 16, (1 << 26) // max(1, index.num_colors())))
        batches = FastxReader(a.query_filename, batch=batch, copy=False)
        emitter = KmerEmitter(index, tool)
    except RuntimeError as e:
        print(str(e), file=sys.stderr)
        return 1
    t0 = time.time()
    n = 0
    try:
        out = open(a.output_filename, "wb")
    except OSError:
        print("could not open output file " + a.output_filename, file=sys.stderr)
        return 1
    with out:
        while True:
            pb, po, cnt = batches.next_raw()
            if cnt == 0:
                break
            pn, pno = batches.names_raw()
            out.flush()
            emitter.write(pb, po, pn, pno, cnt, out.fileno())
            n += cnt
    emitter.close()
    batches.close()
    el = (time.time() - t0) * 1000.0
    if a.verbose:
        print("processed %d reads" % n)
        print("elapsed = %d millisec / %d sec / %d min / %g musec/read" % (el, el / 1000, el / 60000, el * 1000 / max(1, n)))
    return 0


def dump(argv):
    """`fulgor dump -i index -o basename` (tools/util.cpp): the index as the reference's text interchange files"""
    ap = argparse.ArgumentParser(prog="fulgor dump", add_help=True)
    ap.add_argument("-i", dest="index_filename", required=True)
    ap.add_argument("-o", dest="basename", required=True)
    try:
        a = ap.parse_args(argv)
        Index(a.index_filename, device=-1).dump(a.basename)
    except SystemExit:
        return 1
    except RuntimeError as e:
        print(str(e), file=sys.stderr)
        return 1
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    tools = {"pseudoalign": pseudoalign, "dump": dump,
             "kmer-conservation": lambda av: _query_tool(av, "kmer-conservation", 0),
             "kmer-matches": lambda av: _query_tool(av, "kmer-matches", 1)}
    if not argv or argv[0] not in tools:
        print("usage: python -m fulgor_amd <pseudoalign|kmer-conservation|kmer-matches|dump> -i <index> -q <reads> -o <out> "
              "[-r tau] [--format ascii|binary|compressed] [--deduplicate] [--verbose] [--gpus N] [-t io threads]")
        return 1
    return tools[argv[0]](argv[1:])
