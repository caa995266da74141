"""Read sources for the driver: FASTA/FASTQ parsing (src/ps_utils.cpp:245-305: read id = 0-based file
order) and the seeded synthetic read generator of SURVEY §8(d) (native, csrc/tools/readgen.cpp)."""
import ctypes as C
import gzip
import os

import numpy as np

from . import _build

_tools = None


def _lib():
    global _tools
    if _tools is None:
        if not os.path.exists(_build.LIB_TOOLS):
            raise RuntimeError("%s is missing: run __graft_entry__.build()" % _build.LIB_TOOLS)
        L = C.CDLL(_build.LIB_TOOLS)
        L.fgt_genomes_new.restype = C.c_void_p
        L.fgt_genomes_add_fasta.argtypes = [C.c_void_p, C.c_char_p]
        L.fgt_genomes_add_fasta.restype = C.c_int64
        L.fgt_genomes_add_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.fgt_genomes_add_raw.restype = C.c_int64
        L.fgt_genomes_free.argtypes = [C.c_void_p]
        L.fgt_genomes_free.restype = None
        L.fgt_generate_reads.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
        _tools = L
    return _tools


class ReadGenerator:
    def __init__(self, fasta_paths=(), raw_sequences=()):
        self._L = _lib()
        self._h = C.c_void_p(self._L.fgt_genomes_new())
        for p in fasta_paths:
            if self._L.fgt_genomes_add_fasta(self._h, str(p).encode()) < 0:
                raise RuntimeError("cannot read genome %s" % p)
        for s in raw_sequences:
            a = np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else np.ascontiguousarray(s, dtype=np.uint8)
            self._L.fgt_genomes_add_raw(self._h, a.ctypes.data_as(C.c_void_p), len(a))

    def generate(self, first, count, read_len=150, seed=42, threads=None):
        """reads [first, first+count) of the global read set -> (uint8 bases, uint64 offsets)"""
        out = np.empty(count * read_len, dtype=np.uint8)
        threads = threads or os.cpu_count() or 1
        rc = self._L.fgt_generate_reads(self._h, first, count, read_len, seed, out.ctypes.data_as(C.c_void_p), threads)
        if rc != 0:
            raise RuntimeError("read generation failed (no window of %d ACGT bases?)" % read_len)
        offs = np.arange(count + 1, dtype=np.uint64) * np.uint64(read_len)
        return out, offs

    def close(self):
        if self._h:
            self._L.fgt_genomes_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def parse_fastx(path):
    """FASTA/FASTQ (optionally .gz) -> list of sequences (bytes) in file order."""
    op = gzip.open if str(path).endswith(".gz") else open
    seqs = []
    with op(path, "rb") as f:
        first = f.read(1)
        f.seek(0)
        if first == b"@":
            for i, line in enumerate(f):
                if i % 4 == 1:
                    seqs.append(line.strip())
        else:
            cur = None
            for line in f:
                if line.startswith(b">"):
                    if cur is not None:
                        seqs.append(b"".join(cur))
                    cur = []
                elif cur is not None:
                    cur.append(line.strip())
            if cur is not None:
                seqs.append(b"".join(cur))
    return seqs


ALL = (1 << 64) - 1


def count_reads(path, begin=0, end=ALL, threads=0):
    """number of records that start in the byte range [begin, end) of a plain FASTA/FASTQ file (fgpu_fastx_count)"""
    import ctypes as C
    from . import _native
    n = C.c_uint64()
    _native.check(_native.lib().fgpu_fastx_count(str(path).encode(), int(threads), int(begin), int(end), C.byref(n)))
    return n.value


def text_size(path):
    """(length of the text that parts [begin, end) of the query file refer to, whether it can be read in parts): the file size
    for a plain file, the inflated size for a block-compressed gzip file, (0, False) for an ordinary gzip stream"""
    import ctypes as C
    from . import _native
    n, ok = C.c_uint64(), C.c_int()
    _native.check(_native.lib().fgpu_fastx_text_size(str(path).encode(), C.byref(n), C.byref(ok)))
    return n.value, bool(ok.value)


def is_gzip(path):
    with open(path, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


class FastxReader:
    """native FASTA/FASTQ(.gz) reader (fgpu_fastx_*): plain files are parsed by `threads` threads at once, a gzip stream by
    one, off the caller's thread in both cases. Iterating yields (bases uint8 array, offsets uint64 array) batches of at
    most `batch` reads in file order. copy=False hands out views of the reader's own pinned, recycled buffers, valid until
    THREE more batches have been requested (what the worker loop wants: batches go straight to the GPU, a few in flight).
    begin / end: only the records starting in that byte range of a plain file (one part per GPU)."""

    def __init__(self, path, batch=1 << 20, copy=True, threads=0, begin=0, end=ALL):
        import ctypes as C
        from . import _native
        self._C, self._N, self._L = C, _native, _native.lib()
        self.batch = int(batch)
        self.copy = bool(copy)
        h = C.c_void_p()
        _native.check(self._L.fgpu_fastx_open_part(str(path).encode(), int(threads), int(begin), int(end), C.byref(h)))
        self._h = h

    def __iter__(self):
        return self

    def __next__(self):
        C = self._C
        pb, po, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._N.check(self._L.fgpu_fastx_next(self._h, self.batch, C.byref(pb), C.byref(po), C.byref(n)))
        self._last_n = n.value
        if n.value == 0:
            raise StopIteration
        if self.copy:
            offs = self._N.copy_array(po, n.value + 1, np.uint64)
            return self._N.copy_array(pb, int(offs[-1]), np.uint8), offs
        offs = np.frombuffer((C.c_uint64 * (n.value + 1)).from_address(po.value), dtype=np.uint64)
        total = int(offs[-1])
        bases = np.frombuffer((C.c_ubyte * max(total, 1)).from_address(pb.value), dtype=np.uint8)[:total]
        return bases, offs

    def count(self):
        """records of this reader's part, counted without consuming it (fgpu_fastx_count_part: a walk over the record grammar on
        the reader's threads, nothing copied)"""
        n = self._C.c_uint64()
        self._N.check(self._L.fgpu_fastx_count_part(self._h, self._C.byref(n)))
        return n.value

    def next_raw(self):
        """the next batch as raw addresses into the reader's own buffers: (bases address, offsets address, n); n = 0 at the end
        (for a caller that hands them straight to another native call)"""
        C = self._C
        pb, po, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._N.check(self._L.fgpu_fastx_next(self._h, self.batch, C.byref(pb), C.byref(po), C.byref(n)))
        self._last_n = n.value
        return pb.value or 0, po.value or 0, n.value

    def names_raw(self):
        """(names address, name offsets address) of the batch returned last: the names concatenated + (n + 1) offsets"""
        C = self._C
        pn, po = C.c_void_p(), C.c_void_p()
        self._N.check(self._L.fgpu_fastx_names(self._h, C.byref(pn), C.byref(po)))
        return pn.value or 0, po.value or 0

    def names(self):
        """names of the records of the batch returned last (header up to the first blank), as a list of str"""
        C = self._C
        pn, po = C.c_void_p(), C.c_void_p()
        self._N.check(self._L.fgpu_fastx_names(self._h, C.byref(pn), C.byref(po)))
        n = self._last_n
        offs = self._N.copy_array(po, n + 1, np.uint64)
        raw = bytes(self._N.copy_array(pn, int(offs[-1]), np.uint8))
        return [raw[int(offs[i]):int(offs[i + 1])].decode("latin-1") for i in range(n)]

    def close(self):
        if self._h:
            self._L.fgpu_fastx_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
