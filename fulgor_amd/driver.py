"""Host driver of the pseudoalignment path: the counterpart of pseudoalign_orchestrator / pseudoalign_worker
(tools/pseudoalign.cpp:12-89) with the per-read loop replaced by batched passes through the C ABI, the
output formatters of src/ps_utils.cpp:48-136, and the multi-GPU sharding (reads are independent units:
contiguous ranges per rank, index replicated, one all-reduce of the hit counters)."""
import numpy as np

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
        b = self._C.string_at(p, n.value)
        self._L.fgpu_free(p)
        return b

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


def pseudoalign_reads(index, bases, offs, algo=FULL_INTERSECTION, threshold=0.0, chunk=1 << 20, first_id=0,
                      sink=None, fmt="ascii"):
    """the worker loop over one read set: upload once, one pass per chunk, format + write each chunk.
    returns (num_reads, num_mapped_reads)"""
    n = len(offs) - 1
    reads = index.upload_reads(bases, offs)
    res = index.new_result()
    mapped = 0
    f = Formatter(fmt, index.num_colors()) if sink is not None else None
    if f is not None:
        sink.write(f.header)
    for a in range(0, n, chunk):
        cnt = min(chunk, n - a)
        index.run(reads, res, algo, threshold, a, cnt)
        _, _, m = res.sizes()
        mapped += m
        if f is not None:
            o, c = res.download()
            sink.write(f.add(first_id + a, o, c))
    if f is not None:
        sink.write(f.finish())
    res.close()
    reads.close()
    return n, mapped
