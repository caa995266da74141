"""Host driver of the pseudoalignment path: the counterpart of pseudoalign_orchestrator / pseudoalign_worker
(tools/pseudoalign.cpp:12-89) with the per-read loop replaced by batched passes through the C ABI, the
output formatters of src/ps_utils.cpp:48-136, and the multi-GPU sharding (reads are independent units:
contiguous ranges per rank, index replicated, one all-reduce of the hit counters)."""
import threading

import numpy as np

from . import _native
from .index import FULL_INTERSECTION, THRESHOLD_UNION, pack_reads  # noqa: F401


def shard_range(n_total, rank, world):
    """contiguous range of read indices owned by `rank`: keeps global id = offset + local index"""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


def hit_vector(offsets, colors, num_colors):
    """host definition of the vector that is all-reduced: hits[c] = #reads whose result contains c,
    followed by {num_reads, num_mapped_reads (ps_utils.cpp:444-447)}"""
    v = np.zeros(num_colors + 2, dtype=np.int64)
    if len(colors):
        v[:num_colors] = np.bincount(colors, minlength=num_colors)
    sizes = np.diff(np.asarray(offsets, dtype=np.int64))
    v[num_colors] = len(sizes)
    v[num_colors + 1] = int((sizes > 0).sum())
    return v


def all_reduce_hits(t):
    """sum the hit vector over all ranks (RCCL when the tensor is on a GPU, gloo on CPU); no-op for 1 rank"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return t


# ---- output formatters (src/ps_utils.cpp:48-243): native, through the C ABI -----------------------------------
FORMATS = {"ascii": 0, "binary": 1, "compressed": 2}


class Formatter:
    """formatter_buffer of one worker: add(first_id, offsets, colours) -> bytes; finish() -> trailing bytes"""

    def __init__(self, fmt, num_colors):
        import ctypes as C
        from . import _native
        self._C, self._N = C, _native
        self._L = _native.lib()
        if fmt not in FORMATS:
            raise ValueError("Unknown output format. Supported formats: ascii, binary, compressed.")
        h, p, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _native.check(self._L.fgpu_formatter_create(FORMATS[fmt], int(num_colors), C.byref(h), C.byref(p), C.byref(n)))
        self._h = h
        self.header = self._take(p, n)

    def _take(self, p, n):
        return self._N.take_bytes(p, n.value)

    def add(self, first_id, offsets, colors):
        C = self._C
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        colors = np.ascontiguousarray(colors, dtype=np.uint32)
        p, n = C.c_void_p(), C.c_uint64()
        self._N.check(self._L.fgpu_formatter_add(self._h, int(first_id), offsets.ctypes.data_as(C.c_void_p),
                                                 colors.ctypes.data_as(C.c_void_p), len(offsets) - 1, C.byref(p), C.byref(n)))
        return self._take(p, n)

    def finish(self):
        C = self._C
        p, n = C.c_void_p(), C.c_uint64()
        self._N.check(self._L.fgpu_formatter_finish(self._h, C.byref(p), C.byref(n)))
        self._h = None
        return self._take(p, n)


def format_ascii(first_id, offsets, colors):
    f = Formatter("ascii", 0)
    return f.add(first_id, offsets, colors) + f.finish()


def format_binary(first_id, offsets, colors):
    f = Formatter("binary", 0)
    return f.add(first_id, offsets, colors) + f.finish()


def deduplicated_full_intersection(index, bases, offs):
    """--deduplicate (tools/pseudoalign.cpp:91-226): fetch the colour-set ids of every read, collapse
    identical id lists, intersect each distinct list once, fan the results back out. Same answers as the
    direct path; the intersection kernel sees only distinct id lists. Returns CSR (offsets, colours)."""
    ido, ids = index.fetch_color_set_ids_batch(bases, offs)
    ido = ido.astype(np.int64)
    n = len(ido) - 1
    lens = np.diff(ido)
    maxlen = int(lens.max()) if n else 0
    if n == 0 or maxlen == 0:
        return np.zeros(n + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32)
    # exact grouping: pad the lists to a rectangle (0xFFFFFFFF is not a valid id) and take unique rows
    pad = np.full((n, maxlen), 0xFFFFFFFF, dtype=np.uint32)
    cols = np.arange(len(ids), dtype=np.int64) - np.repeat(ido[:-1], lens)
    pad[np.repeat(np.arange(n), lens), cols] = ids
    uniq, inv = np.unique(pad, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    ulens = (uniq != 0xFFFFFFFF).sum(axis=1)
    u_off = np.zeros(len(uniq) + 1, dtype=np.uint64)
    u_off[1:] = np.cumsum(ulens)
    u_ids = uniq[uniq != 0xFFFFFFFF]  # row-major order keeps every list contiguous and sorted
    ro, rc = index.intersect_ids_batch(u_ids, u_off)
    ro = ro.astype(np.int64)
    rsz = np.diff(ro)
    sizes = rsz[inv]
    out_off = np.zeros(n + 1, dtype=np.uint64)
    out_off[1:] = np.cumsum(sizes)
    total = int(out_off[-1])
    # gather: element j of read r comes from rc[ro[inv[r]] + j]
    src = np.repeat(ro[inv], sizes) + (np.arange(total, dtype=np.int64) - np.repeat(out_off[:-1].astype(np.int64), sizes))
    return out_off, rc[src] if total else np.zeros(0, dtype=np.uint32)


def pseudoalign_reads(index, bases, offs, algo=FULL_INTERSECTION, threshold=0.0, chunk=1 << 20, first_id=0,
                      sink=None, fmt="ascii", deduplicate=False):
    """the worker loop over one read set: upload once, one pass per chunk, format + write each chunk.
    returns (num_reads, num_mapped_reads)"""
    n = len(offs) - 1
    f = Formatter(fmt, index.num_colors()) if sink is not None else None
    if f is not None:
        sink.write(f.header)
    if deduplicate:
        if algo != FULL_INTERSECTION:
            raise ValueError("Deduplication not available for threshold < 1.0. Remove --deduplicate flag.")
        mapped = 0
        offs = np.asarray(offs, dtype=np.uint64)
        for a in range(0, n, chunk):
            cnt = min(chunk, n - a)
            lo, hi = int(offs[a]), int(offs[a + cnt])
            o, c = deduplicated_full_intersection(index, bases[lo:hi], offs[a:a + cnt + 1] - offs[a])
            mapped += int((np.diff(o.astype(np.int64)) > 0).sum())
            if f is not None:
                sink.write(f.add(first_id + a, o, c))
        if f is not None:
            sink.write(f.finish())
        return n, mapped
    reads = index.upload_reads(bases, offs)
    res = index.new_result()
    mapped = 0
    for a in range(0, n, chunk):
        cnt = min(chunk, n - a)
        index.run(reads, res, algo, threshold, a, cnt)
        _, _, m = res.sizes()
        mapped += m
        if f is not None:  # formatted by HIP kernels from the resident results (fgpu_result_format_view)
            sink.write(res.format_view(FORMATS[fmt], first_id + a))
    if f is not None:
        f.finish()  # (the host formatter only supplied the file header)
    res.close()
    reads.close()
    return n, mapped


def pseudoalign_stream(index, batches, algo=FULL_INTERSECTION, threshold=0.0, sink=None, fmt="ascii", deduplicate=False,
                       first_id=0, write_header=True, inflight=3):
    """the worker loop over a stream of read batches (reads.FastxReader): parsing, upload, kernels, device-side formatting
    and the copy of the formatted records back to the host overlap. `inflight` batches are in the pipeline at a time, each
    driven by its own host thread on its own result (= its own HIP stream; the C calls release the GIL); records are written
    in file order. Read ids follow file order from first_id on. returns (num_reads, num_mapped_reads)"""
    if deduplicate and algo != FULL_INTERSECTION:
        raise ValueError("Deduplication not available for threshold < 1.0. Remove --deduplicate flag.")
    f = Formatter(fmt, index.num_colors()) if sink is not None else None
    if f is not None and write_header:
        sink.write(f.header)
    n = mapped = 0
    if deduplicate:
        for bases, offs in batches:
            o, c = deduplicated_full_intersection(index, bases, offs)
            mapped += int((np.diff(o.astype(np.int64)) > 0).sum())
            if f is not None:
                sink.write(f.add(first_id + n, o, c))
            n += len(offs) - 1
        if f is not None:
            sink.write(f.finish())  # the deduplicated path formats on the host: flush its last compressed block
        return n, mapped
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    # the reader keeps RING batches alive (fgpu_fastx_next): batch k - (RING - 1) must survive the request for batch k
    inflight = max(1, min(_native.lib().fgpu_fastx_ring() - 1, int(inflight)))
    # the results (device buffers sized for one batch, a stream each) stay with the index: a second stream of batches through
    # the same index finds them allocated (allocating and freeing device memory synchronises the whole device). They are
    # checked out of a pool under a lock: two streams running through one index at the same time never share a result.
    lock = index.__dict__.setdefault("_stream_lock", threading.Lock())
    with lock:
        pool_ = index.__dict__.setdefault("_stream_results", [])
        results = [pool_.pop() for _ in range(min(inflight, len(pool_)))]
    while len(results) < inflight:
        results.append(index.new_result())

    def one_pass(slot, bases, offs, id0):
        res = results[slot]
        reads = index.upload_reads(bases, offs)
        index.run(reads, res, algo, threshold)
        m = res.sizes()[2]
        view = res.format_view(FORMATS[fmt], id0) if f is not None else None  # valid until this result formats again
        reads.close()
        return m, view

    pending = deque()

    def retire():
        nonlocal mapped
        m, view = pending.popleft().result()
        mapped += m
        if view is not None:
            sink.write(view)

    try:
        with ThreadPoolExecutor(max_workers=inflight) as pool:
            slot = 0
            for bases, offs in batches:
                if len(pending) == inflight:
                    retire()  # frees the slot that is taken next: slots are used round robin
                pending.append(pool.submit(one_pass, slot, bases, offs, first_id + n))
                slot = (slot + 1) % inflight
                n += len(offs) - 1
            while pending:
                retire()
    finally:
        with lock:
            index._stream_results.extend(results)  # back to the pool
    if f is not None:
        f.finish()  # (the host formatter only supplied the file header)
    return n, mapped


def _mark(what):
    """FULGOR_CLI_TIMELINE=<epoch seconds at which the command was started>: the stages of one command on stderr"""
    import os
    import sys
    import time
    t0 = os.environ.get("FULGOR_CLI_TIMELINE")
    if t0:
        print("[cli] +%.3f s %s" % (time.time() - float(t0), what), file=sys.stderr, flush=True)


def _quiet(fn, *a, **kw):
    """a preparation step that fails leaves the run to prepare itself"""
    try:
        fn(*a, **kw)
    except RuntimeError:
        pass


# ---- several GPUs: one process per GPU, every rank takes one part of the query file --------------------------
def rank_env():
    """(rank, world size, local rank) as torchrun / the CLI's own launcher export them"""
    import os
    rank = int(os.environ.get("RANK", "0"))
    return rank, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", str(rank)))


def reader_threads_per_rank(io_threads, world):
    """parser threads of one rank: what the caller named, else half of the host's hardware threads divided among the ranks that
    share the host (at most 24 per rank, at least 2)"""
    if io_threads:
        return int(io_threads)
    import os
    hw = os.cpu_count() or 2
    return max(2, min(24, hw // (2 * max(1, world))))


def _place_part(part, output, offset):
    """copy the file `part` into `output` at byte `offset` (every rank places its own part: the joins run side by side)"""
    import os
    with open(part, "rb") as src, open(output, "r+b") as dst:
        size = os.fstat(src.fileno()).st_size
        done = 0
        while done < size:
            try:
                k = os.copy_file_range(src.fileno(), dst.fileno(), min(size - done, 1 << 30), done, offset + done)
            except (OSError, AttributeError):
                k = 0
            if k <= 0:  # (file systems without copy_file_range: through user space)
                buf = os.pread(src.fileno(), min(size - done, 64 << 20), done)
                os.pwrite(dst.fileno(), buf, offset + done)
                k = len(buf)
            done += k
    os.remove(part)


def query_head_stats(path, limit=1 << 18):
    """(bytes of text per record, longest sequence, is FASTQ) from the first records of a plain FASTA / FASTQ file; defaults for
    anything else (compressed, empty, odd): what the one-run preparation sizes its buffers by — a wrong guess costs time, nothing else"""
    try:
        with open(path, "rb") as f:
            head = f.read(limit)
        if head[:2] == b"\x1f\x8b" or head[:1] not in (b"@", b">"):
            return 320, 150, True
        fastq = head[:1] == b"@"
        lines = head.split(b"\n")[:-1]
        if fastq:
            seqs = [len(lines[i].rstrip(b"\r")) for i in range(1, len(lines), 4)]
            recs = len(lines) // 4
        else:
            seqs, cur = [], None
            for l in lines:
                if l[:1] == b">":
                    if cur is not None:
                        seqs.append(cur)
                    cur = 0
                elif cur is not None:
                    cur += len(l.rstrip(b"\r"))
            recs = len(seqs)
        if not recs or not seqs:
            return 320, 150, True
        used = sum(len(l) + 1 for l in lines[:recs * 4]) if fastq else len(head)
        return max(1, used // recs), max(seqs), fastq
    except OSError:
        return 320, 150, True


def pseudoalign_sharded(open_index, query, output, algo=FULL_INTERSECTION, threshold=0.0, fmt="ascii", rank=0, world=1,
                        io_threads=0, device_for_reduce=None, batch=0, prepare_device=None):
    """One rank of a multi-GPU `pseudoalign`. Reads are independent units (tools/pseudoalign.cpp:22-51 keeps no state across
    reads but two counters): rank r opens the r-th of `world` byte ranges of the (plain or block-compressed) query file ONCE,
    counts its records by a walk over the record boundaries that copies nothing (fgpu_fastx_count_part), the ranks exchange
    the counts (read ids are file order), and every rank streams its part through the native worker loop
    (fgpu_pseudoalign_stream). Rank 0 writes straight into the output file, the others into <output>.part<r>; the sizes are
    exchanged and every rank places its own part (the joins run side by side); the two counters are all-reduced (RCCL when
    the tensor lives on a GPU, gloo on the CPU). open_index() opens this rank's replica of the index.
    returns (num_reads, num_mapped_reads) of the whole job."""
    import os
    import sys
    from .reads import FastxReader, text_size
    if world > 1:
        import torch
        import torch.distributed as dist
        size, in_parts = text_size(query)
        if not in_parts:
            raise ValueError("the query file cannot be read in parts (a gzip stream, or FASTQ with wrapped lines): for a multi-GPU run "
                             "decompress it or compress it in blocks (bgzip); unwrap the FASTQ records to four lines")
        begin, end = size * rank // world, size * (rank + 1) // world
    else:
        begin, end = 0, (1 << 64) - 1
    prep = None
    if prepare_device is not None:
        # one command = one run of the worker loop: the host buffers it needs are pinned on a thread of their own while the reader
        # starts and the index opens (fgpu_prepare_host; opt-in: a library user that never streams is not charged the pinned memory)
        from .index import prepare_host
        rec_bytes, max_len, fastq = query_head_stats(query)
        # (output records: 256 bytes per read is typical of the compressed format on thousands of colours; ascii / binary records are
        # sized by the first batch)
        try:  # (this rank's share of a plain file; a compressed one is several times its size: unknown)
            with open(query, "rb") as qf:
                plain = qf.read(2) != b"\x1f\x8b"
            part_bytes = (min(end, os.path.getsize(query)) - begin) if plain else 0
        except OSError:
            part_bytes = 0
        if part_bytes > 0 and not batch:  # a query smaller than one default batch: its batch (and what is pinned and allocated for it) is its size
            batch = max(1024, min(1 << 18 if fmt == "compressed" else 1 << 15, part_bytes // max(1, rec_bytes) + 1))
        kw = dict(device=prepare_device, reader_threads=reader_threads_per_rank(io_threads, world), batch=batch, text_bytes_per_read=rec_bytes,
                  fastq=fastq, out_bytes_per_read=256 if fmt == "compressed" else 0, total_text_bytes=max(0, part_bytes))
        prep = threading.Thread(target=lambda: _quiet(prepare_host, **kw))
        prep.start()
    _mark("preparation thread started" if prep is not None else "no preparation thread")
    trace = os.environ.get("FULGOR_TRACE_OPENS") == "1"  # (tests: one open of the query part and one of the index per rank)
    made = {}

    def open_reader():
        """this rank's part of the query file: opened once (its threads start parsing at once) and, for a multi-GPU run, counted"""
        try:
            made["reader"] = FastxReader(query, batch=batch or 1 << 18, copy=False, threads=reader_threads_per_rank(io_threads, world),
                                         begin=begin, end=end)
            _mark("query reader open")
            if trace:
                print("[rank] query part opened (rank %d/%d, text bytes %d..%d)" % (rank, world, begin, min(end, 1 << 62)), file=sys.stderr, flush=True)
            if world > 1:
                made["count"] = made["reader"].count()
        except Exception as e:  # noqa: BLE001 — raised on the caller's thread below
            made["error"] = e

    index = None
    if prepare_device is not None:
        # (the command line: the reader opens on a thread of its own WHILE the index opens — both wait for the HIP runtime to start, a
        # fifth of a second in a fresh process, and the container is read meanwhile)
        th = threading.Thread(target=open_reader)
        th.start()
        try:
            index = open_index()
        finally:
            th.join()
    else:
        open_reader()
    if "error" in made:
        raise made["error"]
    batches = made["reader"]
    first_id = 0
    if world > 1:
        mine = torch.tensor([made["count"]], dtype=torch.int64, device=device_for_reduce)
        counts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(counts, mine)
        first_id = int(sum(int(c.item()) for c in counts[:rank]))
    if index is None:
        index = open_index()
    if prep is not None:
        prep.join()
        _mark("host buffers pinned")
        if hasattr(index, "stream_prepare"):
            _quiet(index.stream_prepare, FORMATS[fmt], batch, 0, max_len, 256 if fmt == "compressed" else 0)
        _mark("worker results created")
    if trace:
        print("[rank] index opened (rank %d/%d)" % (rank, world), file=sys.stderr, flush=True)
    if world > 1 and hasattr(index, "device_report"):  # every rank of a multi-GPU run says where it runs and which copy engines it chose
        print("[rank %d/%d] first read id %d; %s" % (rank, world, first_id, index.device_report()), file=sys.stderr, flush=True)
    part = output if rank == 0 else "%s.part%d" % (output, rank)
    with open(part, "wb") as out:
        if hasattr(index, "pseudoalign_stream"):  # the engine: the native worker loop
            out.flush()
            _mark("stream starts")
            n, mapped = index.pseudoalign_stream(batches, out.fileno(), algo, threshold, FORMATS[fmt], first_id, rank == 0, batch)
            _mark("stream done")
        else:  # (an index that is not the engine's: the CPU tests of this sharding logic)
            n, mapped = pseudoalign_stream(index, batches, algo, threshold, sink=out, fmt=fmt, first_id=first_id,
                                           write_header=rank == 0)
    batches.close()
    _mark("reader closed")
    if world > 1:
        t = torch.tensor([n, mapped], dtype=torch.int64, device=device_for_reduce)
        dist.all_reduce(t)
        n, mapped = int(t[0].item()), int(t[1].item())
        mine = torch.tensor([os.path.getsize(part)], dtype=torch.int64, device=device_for_reduce)
        sizes = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(sizes, mine)  # (every part is complete when this returns)
        if rank > 0:
            _place_part(part, output, int(sum(int(x.item()) for x in sizes[:rank])))
        dist.barrier()
    return n, mapped


# ---- the reference's --deduplicate temp files (tools/pseudoalign.cpp:91-226, src/ps_utils.cpp:307-415) -------
# Stage 1 writes, per read, `u32 read_id, u32 num_ids, u32 x num_ids`; the deduplication pass rewrites the file as
# `u32 list_len, u32 read_id, u32 x (list_len - 1)` sorted by id list, where list_len == 1 marks a read whose
# id list equals the previous record's (preprocessed_query_reader, ps_utils.cpp:327-369). Reads without ids never
# reach the second file: the reference writes their empty result straight to the output (pseudoalign.cpp:186-189).
def write_fetched_ids(path, id_offs, ids, first_read_id=0):
    """stage-1 temp file for reads first_read_id .. first_read_id + n - 1"""
    id_offs = np.asarray(id_offs, dtype=np.int64)
    ids = np.asarray(ids, dtype=np.uint32)
    n = len(id_offs) - 1
    lens = np.diff(id_offs).astype(np.uint32)
    out = np.empty(2 * n + len(ids), dtype=np.uint32)
    pos = 2 * np.arange(n, dtype=np.int64) + id_offs[:-1]
    out[pos] = np.arange(first_read_id, first_read_id + n, dtype=np.uint32)
    out[pos + 1] = lens
    mask = np.ones(len(out), dtype=bool)
    mask[pos] = False
    mask[pos + 1] = False
    out[mask] = ids
    with open(path, "ab") as f:
        f.write(out.tobytes())


def read_fetched_ids(path):
    """-> (read_ids, id_offs, ids) of a stage-1 temp file"""
    raw = np.fromfile(path, dtype=np.uint32)
    rid, offs, chunks, p = [], [0], [], 0
    while p < len(raw):
        rid.append(int(raw[p]))
        m = int(raw[p + 1])
        chunks.append(raw[p + 2:p + 2 + m])
        offs.append(offs[-1] + m)
        p += 2 + m
    ids = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint32)
    return np.asarray(rid, dtype=np.uint32), np.asarray(offs, dtype=np.uint64), ids.astype(np.uint32)


def deduplicate_fetched(read_ids, id_offs, ids):
    """the reference's sort + collapse (pseudoalign.cpp:193-216): -> (unmapped read ids, records) where a record is
    (read_id, id list or None when it repeats the previous record's list), in lexicographic order of the lists"""
    id_offs = np.asarray(id_offs, dtype=np.int64)
    lists = [(tuple(int(x) for x in ids[id_offs[i]:id_offs[i + 1]]), int(read_ids[i])) for i in range(len(read_ids))]
    unmapped = [r for l, r in lists if not l]
    mapped = sorted(((l, r) for l, r in lists if l), key=lambda t: t[0])  # stable, like std::sort on equal keys modulo order
    records, prev = [], None
    for l, r in mapped:
        records.append((r, None if l == prev else l))
        prev = l
    return unmapped, records


def write_preprocessed(path, records):
    with open(path, "wb") as f:
        for r, l in records:
            body = [r] + (list(l) if l is not None else [])
            f.write(np.asarray([len(body)] + body, dtype=np.uint32).tobytes())


def read_preprocessed(path, batch=10000):
    """preprocessed_query_reader: yields batches of (read_id, id list); a list_len == 1 record repeats the previous list"""
    raw = np.fromfile(path, dtype=np.uint32)
    out, prev, p = [], None, 0
    while p < len(raw):
        s = int(raw[p])
        rid = int(raw[p + 1])
        if s > 1:
            prev = raw[p + 2:p + 1 + s].astype(np.uint32)
        elif prev is None:
            raise ValueError("preprocessed query file starts with a duplicate marker")
        out.append((rid, prev))
        p += 1 + s
        if len(out) == batch:
            yield out
            out = []
    if out:
        yield out


def intersect_preprocessed(index, path, batch=10000):
    """stage 2 of --deduplicate over a preprocessed query file: every distinct id list is intersected once
    (fgpu_intersect_ids), duplicates reuse the previous result. Yields (read_id, colours) in file order."""
    for recs in read_preprocessed(path, batch):
        heads = [i for i, (_, l) in enumerate(recs) if i == 0 or l is not recs[i - 1][1]]
        lists = [recs[i][1] for i in heads]
        ido = np.zeros(len(lists) + 1, dtype=np.uint64)
        ido[1:] = np.cumsum([len(l) for l in lists])
        ro, rc = index.intersect_ids_batch(np.concatenate(lists).astype(np.uint32), ido)
        ro = ro.astype(np.int64)
        h = -1
        for i, (rid, _) in enumerate(recs):
            if h + 1 < len(heads) and heads[h + 1] == i:
                h += 1
            yield rid, rc[ro[h]:ro[h + 1]]
