"""Host-side mirror of the reference's index API for the pseudoalignment path.

Same member names and argument meaning as `template <typename ColorSets> struct index`
(include/index.hpp:16-110 in jermp/fulgor): `fetch_color_set_ids`, `pseudoalign_full_intersection`,
`pseudoalign_threshold_union`, `k`, `num_colors`, ... The per-read members are thin single-read calls;
the `*_batch` members are what a driver loop uses (one C-ABI call per batch of reads). Every call runs
the HIP kernels through libfulgor_gpu.so.
"""
import ctypes as C

import numpy as np

from . import _native

FULL_INTERSECTION = 0
THRESHOLD_UNION = 1
HYBRID, DIFF, META, META_DIFF = 0, 1, 2, 3  # index_t, include/util.hpp:18
KERNELS = ("k1_lookup", "k2_intersect", "k3_union", "scan", "k2b_expand", "k_hits", "k_desc", "k_format", "k_order", "h2d", "d2h")


def pack_reads(reads):
    """list of str/bytes -> (uint8 bases, uint64 offsets[n+1])"""
    bs = [r if isinstance(r, (bytes, bytearray)) else r.encode("ascii") for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(bases), offs


def conservation_triples(kmer_ids):
    """run-length encoding of per-k-mer colour-set ids; negative k-mers (0xFFFFFFFF) break runs"""
    ids = np.asarray(kmer_ids, dtype=np.uint32)
    if len(ids) == 0:
        return []
    change = np.flatnonzero(np.concatenate(([True], ids[1:] != ids[:-1])))
    ends = np.concatenate((change[1:], [len(ids)]))
    return [(int(a), int(b - a), int(ids[a])) for a, b in zip(change, ends) if ids[a] != 0xFFFFFFFF]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _take_csr(L, n, p_off, p_val):
    offs = _native.copy_array(p_off, n + 1, np.uint64)
    total = int(offs[n])
    vals = _native.copy_array(p_val, total, np.uint32)
    L.fgpu_free(p_off)
    L.fgpu_free(p_val)
    return offs, vals


def prepare_host(device=0, reader_threads=0, workers=0, batch=0, text_bytes_per_read=320, fastq=True, out_bytes_per_read=0, total_text_bytes=0):
    """opt-in, needs no index (fgpu_prepare_host): starts the HIP runtime on `device` and pins the host buffers one run of the streamed
    worker loop will use into the process-wide pool. The command line calls it on a thread of its own while the index opens."""
    _native.check(_native.lib().fgpu_prepare_host(int(device), int(reader_threads), int(workers), int(batch), int(text_bytes_per_read),
                                                  1 if fastq else 0, int(out_bytes_per_read), int(total_text_bytes)))


class Reads:
    """A batch of reads resident in HBM."""

    def __init__(self, index, bases, offs):
        self._L = _native.lib()
        self.index = index
        self.n = len(offs) - 1
        h = C.c_void_p()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        _native.check(self._L.fgpu_reads_upload(index._h, _ptr(bases), _ptr(offs), self.n, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            self._L.fgpu_reads_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Result:
    """CSR results of one pass, resident in HBM; reusable across passes."""

    def __init__(self, index):
        self._L = _native.lib()
        self.index = index
        h = C.c_void_p()
        _native.check(self._L.fgpu_result_create(index._h, C.byref(h)))
        self._h = h

    def sizes(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _native.check(self._L.fgpu_result_sizes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value  # reads, total colours, mapped reads

    def expand(self):
        """materialise the u32 colour lists (CSR) of the last pass on the device now (k2b_expand); download(), ascii and binary
        formatting do it on demand, the compressed format and the counters never need them"""
        _native.check(self._L.fgpu_result_expand(self._h))

    def download(self):
        n, total, _ = self.sizes()
        offs = np.zeros(n + 1, dtype=np.uint64)
        cols = np.zeros(max(total, 1), dtype=np.uint32)
        _native.check(self._L.fgpu_result_download(self._h, _ptr(offs), _ptr(cols)))
        return offs, cols[:total]

    def format(self, fmt, first_read_id=0):
        """the records of this pass as the reference writes them (0 = ascii, 1 = binary), formatted on the device"""
        p, n = C.c_void_p(), C.c_uint64()
        _native.check(self._L.fgpu_result_format(self._h, int(fmt), int(first_read_id), C.byref(p), C.byref(n)))
        return _native.take_bytes(p, n.value)

    def format_view(self, fmt, first_read_id=0):
        """like format(), without copies: a memoryview of the result's pinned host buffer, valid until the next
        format call on this result (write it to the output file, then move on)"""
        p, n = C.c_void_p(), C.c_uint64()
        _native.check(self._L.fgpu_result_format_view(self._h, int(fmt), int(first_read_id), C.byref(p), C.byref(n)))
        if n.value == 0:
            return memoryview(b"")
        return memoryview((C.c_char * n.value).from_address(p.value)).cast("B")

    def accumulate_hits(self, device_ptr):
        _native.check(self._L.fgpu_result_accumulate_hits(self.index._h, self._h, C.c_void_p(device_ptr)))

    def algorithmic_bytes(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _native.check(self._L.fgpu_result_algorithmic_bytes(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"lists": a.value, "output": b.value, "lookup": c.value}

    def distinct_lists(self):
        """distinct id lists of the last pass under tune(deduplicate=True) (0: the pass was not deduplicated)"""
        n = C.c_uint64()
        _native.check(self._L.fgpu_result_distinct_lists(self._h, C.byref(n)))
        return n.value

    def checksum(self):
        """(from the lists, from the rows): two independent device-side checksums of the u32 colour lists of the last pass —
        {#entries, sum, xor} each; equal when the expansion kernel wrote every colour at its place (fgpu_result_checksum)"""
        a, b = (C.c_uint64 * 3)(), (C.c_uint64 * 3)()
        _native.check(self._L.fgpu_result_checksum(self._h, a, b))
        return tuple(a), tuple(b)

    def close(self):
        if self._h:
            self._L.fgpu_result_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class KmerEmitter:
    """one worker of `fulgor kmer-conservation` (tool 0) or `fulgor kmer-matches` (tool 1) as a native line emitter
    (fgpu_kmer_emitter_*): add(reader) takes the reader's current batch and returns the tool's output lines for it"""

    def __init__(self, index, tool):
        self._L = _native.lib()
        self.index = index
        h = C.c_void_p()
        _native.check(self._L.fgpu_kmer_emitter_create(index._h, int(tool), C.byref(h)))
        self._h = h

    def add(self, bases, offs, names, name_offs, n):
        """bases / offs / names / name_offs: numpy arrays or raw addresses; n records -> bytes"""
        def addr(x):
            return C.c_void_p(x) if isinstance(x, int) else _ptr(x)
        p, ln = C.c_void_p(), C.c_uint64()
        _native.check(self._L.fgpu_kmer_emitter_add(self._h, addr(bases), addr(offs), int(n), addr(names), addr(name_offs), C.byref(p), C.byref(ln)))
        return _native.take_bytes(p, ln.value)

    def write(self, bases, offs, names, name_offs, n, out_fd):
        """as add(), the lines written to the file descriptor out_fd; returns the number of bytes"""
        def addr(x):
            return C.c_void_p(x) if isinstance(x, int) else _ptr(x)
        ln = C.c_uint64()
        _native.check(self._L.fgpu_kmer_emitter_write(self._h, addr(bases), addr(offs), int(n), addr(names), addr(name_offs), int(out_fd), C.byref(ln)))
        return ln.value

    def close(self):
        if self._h:
            self._L.fgpu_kmer_emitter_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Index:
    """index<ColorSets> resident in HBM (load = essentials::load + one upload)."""

    def __init__(self, path, device=0):
        self._L = _native.lib()
        h = C.c_void_p()
        _native.check(self._L.fgpu_open(str(path).encode(), int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        vals = [C.c_uint64() for _ in range(5)]
        t = C.c_int()
        _native.check(self._L.fgpu_info(h, *[C.byref(v) for v in vals], C.byref(t)))
        self._k, self._num_colors, self._num_color_sets, self._num_unitigs, self._num_kmers = [v.value for v in vals]
        self.index_type = t.value

    # --- accessors, include/index.hpp:64-68 ---
    def k(self):
        return self._k

    def num_colors(self):
        return self._num_colors

    def num_color_sets(self):
        return self._num_color_sets

    def num_unitigs(self):
        return self._num_unitigs

    def num_kmers(self):
        return self._num_kmers

    def save(self, path):
        _native.check(self._L.fgpu_save(self._h, str(path).encode()))

    def convert(self, index_type, partition_size=128, cluster_size=16):
        """re-encode the colour sets: 0 hybrid, 1 differential, 2 meta, 3 meta-differential (index_t, util.hpp:18)"""
        _native.check(self._L.fgpu_convert(self._h, int(index_type), int(partition_size), int(cluster_size)))
        self.index_type = int(index_type)
        return self

    def selfcheck(self, unitig_stride=1):
        _native.check(self._L.fgpu_selfcheck(self._h, unitig_stride))

    def close(self):
        if getattr(self, "_h", None):
            for r in getattr(self, "_stream_results", []):  # (driver.pseudoalign_stream keeps its results with the index)
                r.close()
            self._stream_results = []
            self._L.fgpu_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # --- batch members (one C-ABI call each) ---
    def _call(self, fn, bases, offs, *extra):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        po, pv = C.c_void_p(), C.c_void_p()
        _native.check(fn(self._h, _ptr(bases), _ptr(offs), n, *extra, C.byref(po), C.byref(pv)))
        return _take_csr(self._L, n, po, pv)

    def fetch_color_set_ids_batch(self, bases, offs):
        return self._call(self._L.fgpu_fetch_color_set_ids, bases, offs)

    def pseudoalign_full_intersection_batch(self, bases, offs):
        return self._call(self._L.fgpu_full_intersection, bases, offs)

    def pseudoalign_threshold_union_batch(self, bases, offs, threshold):
        return self._call(self._L.fgpu_threshold_union, bases, offs, C.c_double(threshold))

    def kmer_color_set_ids_batch(self, bases, offs):
        """colour-set id of every k-mer of every read (0xFFFFFFFF = negative) as CSR"""
        return self._call(self._L.fgpu_kmer_color_set_ids, bases, offs)

    def kmer_matches_batch(self, bases, offs):
        """index::kmer_matches for a batch: (positive k-mer flags as CSR of 0/1, counts[n_reads, num_colors])"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        p = C.c_void_p()
        _native.check(self._L.fgpu_kmer_matches(self._h, _ptr(bases), _ptr(offs), n, C.byref(p)))
        nc = self._num_colors
        counts = _native.copy_array(p, n * nc, np.uint32).reshape(n, nc)
        self._L.fgpu_free(p)
        ko, ki = self.kmer_color_set_ids_batch(bases, offs)
        return ko, (ki != 0xFFFFFFFF).astype(np.uint8), counts

    def kmer_conservation(self, sequence):
        """index::kmer_conservation (src/kmer_conservation.cpp:7-54): list of (start_pos_in_query, num_kmers, color_set_id)"""
        b, o = pack_reads([sequence])
        _, ids = self.kmer_color_set_ids_batch(b, o)
        return conservation_triples(ids)

    def intersect_ids_batch(self, ids, id_offs):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        id_offs = np.ascontiguousarray(id_offs, dtype=np.uint64)
        n = len(id_offs) - 1
        po, pv = C.c_void_p(), C.c_void_p()
        _native.check(self._L.fgpu_intersect_ids(self._h, _ptr(ids), _ptr(id_offs), n, C.byref(po), C.byref(pv)))
        return _take_csr(self._L, n, po, pv)

    # --- per-read members with the reference's names ---
    def fetch_color_set_ids(self, sequence):
        b, o = pack_reads([sequence])
        return self.fetch_color_set_ids_batch(b, o)[1].tolist()

    def pseudoalign_full_intersection(self, color_set_ids):
        ids = np.asarray(color_set_ids, dtype=np.uint32)
        return self.intersect_ids_batch(ids, np.array([0, len(ids)], dtype=np.uint64))[1].tolist()

    def pseudoalign_threshold_union(self, sequence, threshold):
        b, o = pack_reads([sequence])
        return self.pseudoalign_threshold_union_batch(b, o, threshold)[1].tolist()

    # --- device-resident driver API ---
    def upload_reads(self, bases, offs):
        return Reads(self, bases, offs)

    def new_result(self):
        return Result(self)

    def run(self, reads, result, algo=FULL_INTERSECTION, threshold=0.0, first=0, count=None):
        if count is None:
            count = reads.n - first
        _native.check(self._L.fgpu_run(self._h, reads._h, first, count, algo, C.c_double(threshold), result._h))

    def run_lookup(self, reads, result, first=0, count=None):
        """first half of run(): queues the lookup of the reads on the result's lookup stream and returns at once"""
        if count is None:
            count = reads.n - first
        _native.check(self._L.fgpu_run_lookup(self._h, reads._h, first, count, result._h))

    def run_colours(self, result, algo=FULL_INTERSECTION, threshold=0.0):
        """second half of run(): the colour stage over the ids the result holds; returns when its kernels have completed"""
        _native.check(self._L.fgpu_run_colours(self._h, algo, C.c_double(threshold), result._h))

    def pseudoalign_stream(self, reader, out_fd, algo=FULL_INTERSECTION, threshold=0.0, fmt=0, first_read_id=0, write_header=True,
                           batch=0, workers=0):
        """the native worker loop (fgpu_pseudoalign_stream; pseudoalign_orchestrator of tools/pseudoalign.cpp:53-89): every record
        of `reader` (reads.FastxReader, consumed) -> records in format `fmt` on the file descriptor out_fd, in file order.
        returns (num_reads, num_mapped_reads)"""
        n, m = C.c_uint64(), C.c_uint64()
        _native.check(self._L.fgpu_pseudoalign_stream(self._h, reader._h, int(out_fd), int(algo), C.c_double(threshold), int(fmt),
                                                      int(first_read_id), 1 if write_header else 0, int(batch), int(workers),
                                                      C.byref(n), C.byref(m)))
        return n.value, m.value

    def stream_prepare(self, fmt=0, batch=0, workers=0, max_read_bases=150, out_bytes_per_read=0):
        """opt-in, for a process that streams once (the command line): creates the worker loop's results — streams, device buffers
        sized for its batches, pinned output buffers — ahead of pseudoalign_stream (fgpu_stream_prepare)"""
        _native.check(self._L.fgpu_stream_prepare(self._h, int(fmt), int(batch), int(workers), int(max_read_bases), int(out_bytes_per_read)))

    def last_stream_report(self):
        """timeline of the last pseudoalign_stream of this process (text)"""
        p = C.c_void_p()
        _native.check(self._L.fgpu_last_stream_report(C.byref(p)))
        try:
            return C.string_at(p.value).decode()
        finally:
            self._L.fgpu_free(p)

    def device_report(self):
        """one line about the device of this handle and the copy engines the library chose (fgpu_device_report)"""
        p = C.c_void_p()
        _native.check(self._L.fgpu_device_report(self._h, C.byref(p)))
        try:
            return C.string_at(p.value).decode()
        finally:
            self._L.fgpu_free(p)

    def tune(self, order_min_reads=None, small_results=None, dense_rows=None, deduplicate=None):
        """execution knobs of the colour stage (fgpu_tune); results never depend on them. deduplicate: `--deduplicate` on the
        device (every distinct id list of a pass intersected once)"""
        if deduplicate is not None:
            _native.check(self._L.fgpu_tune(self._h, 3, 1 if deduplicate else 0))
        if order_min_reads is not None:
            _native.check(self._L.fgpu_tune(self._h, 0, int(order_min_reads) if order_min_reads >= 0 else 0xFFFFFFFFFFFFFFFF))
        if small_results is not None:
            _native.check(self._L.fgpu_tune(self._h, 1, 1 if small_results else 0))
        if dense_rows is not None:
            _native.check(self._L.fgpu_tune(self._h, 2, 1 if dense_rows else 0))
        return self

    def timing_enable(self, on=True):
        _native.check(self._L.fgpu_timing_enable(self._h, 1 if on else 0))

    def timing_reset(self):
        _native.check(self._L.fgpu_timing_reset(self._h))

    def timing(self):
        out = {}
        for i, name in enumerate(KERNELS):
            ms, n = C.c_double(), C.c_uint64()
            _native.check(self._L.fgpu_timing_get(self._h, i, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def dump(self, basename):
        """`fulgor dump`: the index as the reference's four text files (src/index.cpp:59-120)"""
        _native.check(self._L.fgpu_dump(self._h, str(basename).encode()))

    def export(self):
        """Encoded index content (unitigs + hybrid colour stream) as numpy arrays."""
        v = [C.c_uint64() for _ in range(5)]
        _native.check(self._L.fgpu_export_sizes(self._h, *[C.byref(x) for x in v]))
        nb, nu, nw, nbits, ns = [x.value for x in v]
        out = {
            "k": self._k,
            "unitig_bases": np.zeros(nb, dtype=np.uint8),
            "unitig_off": np.zeros(nu + 1, dtype=np.uint64),
            "unitig_csid": np.zeros(nu, dtype=np.uint32),
            "color_words": np.zeros(nw, dtype=np.uint64),
            "color_offsets": np.zeros(ns + 1, dtype=np.uint64),
            "thresholds": np.zeros(3, dtype=np.uint32),
            "color_bits": nbits,
        }
        _native.check(self._L.fgpu_export(self._h, _ptr(out["unitig_bases"]), _ptr(out["unitig_off"]),
                                          _ptr(out["unitig_csid"]), _ptr(out["color_words"]),
                                          _ptr(out["color_offsets"]), _ptr(out["thresholds"])))
        return out
