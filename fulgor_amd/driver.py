"""Host driver of the pseudoalignment path: the counterpart of pseudoalign_orchestrator / pseudoalign_worker
(tools/pseudoalign.cpp:12-89) with the per-read loop replaced by batched passes through the C ABI, the
output formatters of src/ps_utils.cpp:48-136, and the multi-GPU sharding (reads are independent units:
contiguous ranges per rank, index replicated, one all-reduce of the hit counters)."""
import struct

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


# ---- output formatters (src/ps_utils.cpp:48-136) -------------------------------------------------------
def format_ascii(first_id, offsets, colors):
    """psa_ascii_formatter: "<id>\\t<count>[\\t<colour>...]\\n" for every read, mapped or not"""
    offsets = np.asarray(offsets, dtype=np.int64)
    out = []
    cs = colors.astype(str) if len(colors) else colors
    for i in range(len(offsets) - 1):
        a, b = offsets[i], offsets[i + 1]
        if b > a:
            out.append("%d\t%d\t%s\n" % (first_id + i, b - a, "\t".join(cs[a:b])))
        else:
            out.append("%d\t0\n" % (first_id + i))
    return "".join(out).encode()


def format_binary(first_id, offsets, colors):
    """psa_binary_formatter: u32 id, u32 count, u32 x count"""
    offsets = np.asarray(offsets, dtype=np.int64)
    out = []
    for i in range(len(offsets) - 1):
        a, b = offsets[i], offsets[i + 1]
        out.append(struct.pack("<II", first_id + i, b - a))
        out.append(colors[a:b].astype("<u4").tobytes())
    return b"".join(out)


FORMATTERS = {"ascii": format_ascii, "binary": format_binary}


def pseudoalign_reads(index, bases, offs, algo=FULL_INTERSECTION, threshold=0.0, chunk=1 << 20, first_id=0,
                      sink=None, fmt="ascii"):
    """the worker loop over one read set: upload once, one pass per chunk, format + write each chunk.
    returns (num_reads, num_mapped_reads)"""
    n = len(offs) - 1
    reads = index.upload_reads(bases, offs)
    res = index.new_result()
    mapped = 0
    for a in range(0, n, chunk):
        cnt = min(chunk, n - a)
        index.run(reads, res, algo, threshold, a, cnt)
        _, _, m = res.sizes()
        mapped += m
        if sink is not None:
            o, c = res.download()
            sink.write(FORMATTERS[fmt](first_id + a, o, c))
    res.close()
    reads.close()
    return n, mapped
