"""ORACLE — test infrastructure only (PARITY UNPINNED, see fulgor_oracle.hpp). ctypes wrapper over
liboracle.so, the CPU restatement of the reference's pseudoalignment path. Importable only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle`")
        L = C.CDLL(LIB)
        vp = C.c_void_p
        L.fo_last_error.restype = C.c_char_p
        L.fo_index_load_dump.restype = vp
        L.fo_index_load_dump.argtypes = [C.c_char_p]
        L.fo_index_from_arrays.restype = vp
        L.fo_index_from_arrays.argtypes = [C.c_uint32, vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                           vp, C.c_uint64, vp, C.c_uint64]
        L.fo_index_free.argtypes = [vp]
        L.fo_index_free.restype = None
        L.fo_index_info.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 5
        L.fo_index_info.restype = None
        L.fo_colors_words.restype = C.POINTER(C.c_uint64)
        L.fo_colors_words.argtypes = [vp]
        L.fo_colors_offsets.restype = C.POINTER(C.c_uint64)
        L.fo_colors_offsets.argtypes = [vp]
        L.fo_index_convert.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32]
        L.fo_free.argtypes = [vp]
        L.fo_free.restype = None
        L.fo_fetch_color_set_ids.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.c_int]
        L.fo_full_intersection.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int]
        L.fo_intersect_ids.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int]
        L.fo_threshold_union.argtypes = [vp, vp, vp, C.c_uint64, C.c_double, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int]
        L.fo_time_pseudoalign.restype = C.c_double
        L.fo_time_pseudoalign.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, C.c_double, C.c_int,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.fo_format_compressed.restype = vp
        L.fo_format_compressed.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        L.fo_parse_compressed.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.fo_kmer_conservation.argtypes = [vp, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(vp)]
        L.fo_kmer_matches.argtypes = [vp, C.c_char_p, C.c_uint64, vp, vp]
        L.fo_format_ascii.restype = vp
        L.fo_format_ascii.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _take(L, n, po, pv):
    offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    total = int(offs[n])
    vals = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_uint32)), shape=(max(total, 1),))[:total].copy()
    L.fo_free(po)
    L.fo_free(pv)
    return offs, vals


def format_compressed(offs, colors, num_colors, first_id=0):
    """psa_compressed_formatter of one worker (src/ps_utils.cpp:138-243) over a whole CSR result"""
    L = lib()
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    colors = np.ascontiguousarray(colors, dtype=np.uint32)
    ln = C.c_uint64()
    p = L.fo_format_compressed(_ptr(offs), _ptr(colors), len(offs) - 1, first_id, num_colors, C.byref(ln))
    s = C.string_at(p, ln.value)
    L.fo_free(p)
    return s


def parse_compressed(data):
    """inverse of format_compressed -> (ids, offsets, colours)"""
    L = lib()
    n = C.c_uint64()
    pi, po, pc = C.c_void_p(), C.c_void_p(), C.c_void_p()
    if L.fo_parse_compressed(data, len(data), C.byref(n), C.byref(pi), C.byref(po), C.byref(pc)) != 0:
        raise RuntimeError("oracle: %s" % L.fo_last_error().decode())
    ids = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint32)), shape=(max(n.value, 1),))[:n.value].copy()
    offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n.value + 1,)).copy()
    tot = int(offs[-1])
    cols = np.ctypeslib.as_array(C.cast(pc, C.POINTER(C.c_uint32)), shape=(max(tot, 1),))[:tot].copy()
    for q in (pi, po, pc):
        L.fo_free(q)
    return ids, offs, cols


class OracleIndex:
    def __init__(self, handle):
        self._L = lib()
        if not handle:
            raise RuntimeError("oracle: %s" % self._L.fo_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def from_dump(cls, base):
        return cls(lib().fo_index_load_dump(str(base).encode()))

    @classmethod
    def from_export(cls, ex):
        """ex = fulgor_amd.Index.export(): same encoded colour stream and unitigs the GPU holds"""
        t = ex["thresholds"]
        return cls(lib().fo_index_from_arrays(
            int(ex["k"]), _ptr(ex["unitig_bases"]), _ptr(ex["unitig_off"]), _ptr(ex["unitig_csid"]),
            len(ex["unitig_csid"]), int(t[0]), int(t[1]), int(t[2]), _ptr(ex["color_words"]), int(ex["color_bits"]),
            _ptr(ex["color_offsets"]), len(ex["color_offsets"]) - 1))

    def convert(self, index_type, partition_size=64, cluster_size=8):
        """re-encode the colour sets: 0 hybrid, 1 differential, 2 meta, 3 meta-differential"""
        if self._L.fo_index_convert(self._h, index_type, partition_size, cluster_size) != 0:
            raise RuntimeError("oracle: %s" % self._L.fo_last_error().decode())
        return self

    def info(self):
        v = [C.c_uint64() for _ in range(5)]
        self._L.fo_index_info(self._h, *[C.byref(x) for x in v])
        return dict(zip(("k", "num_colors", "num_sets", "num_unitigs", "color_bits"), [x.value for x in v]))

    def encoded_colors(self):
        i = self.info()
        nw = (i["color_bits"] + 63) // 64
        words = np.ctypeslib.as_array(self._L.fo_colors_words(self._h), shape=(max(nw, 1),))[:nw].copy()
        offs = np.ctypeslib.as_array(self._L.fo_colors_offsets(self._h), shape=(i["num_sets"] + 1,)).copy()
        return words, offs

    def _run(self, fn, bases, offs, *extra):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        po, pv = C.c_void_p(), C.c_void_p()
        if fn(self._h, _ptr(bases), _ptr(offs), n, *extra[:1], C.byref(po), C.byref(pv), *extra[1:]) != 0:
            raise RuntimeError("oracle: %s" % self._L.fo_last_error().decode())
        return _take(self._L, n, po, pv)

    def fetch_color_set_ids(self, bases, offs, threads=8):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        po, pv = C.c_void_p(), C.c_void_p()
        if self._L.fo_fetch_color_set_ids(self._h, _ptr(bases), _ptr(offs), n, C.byref(po), C.byref(pv), threads) != 0:
            raise RuntimeError("oracle: %s" % self._L.fo_last_error().decode())
        return _take(self._L, n, po, pv)

    def full_intersection(self, bases, offs, threads=8, self_check=False):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        po, pv = C.c_void_p(), C.c_void_p()
        if self._L.fo_full_intersection(self._h, _ptr(bases), _ptr(offs), n, C.byref(po), C.byref(pv), threads,
                                        int(self_check)) != 0:
            raise RuntimeError("oracle: %s" % self._L.fo_last_error().decode())
        return _take(self._L, n, po, pv)

    def intersect_ids(self, ids, id_offs, threads=8, self_check=False):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        id_offs = np.ascontiguousarray(id_offs, dtype=np.uint64)
        n = len(id_offs) - 1
        po, pv = C.c_void_p(), C.c_void_p()
        if self._L.fo_intersect_ids(self._h, _ptr(ids), _ptr(id_offs), n, C.byref(po), C.byref(pv), threads,
                                    int(self_check)) != 0:
            raise RuntimeError("oracle: %s" % self._L.fo_last_error().decode())
        return _take(self._L, n, po, pv)

    def threshold_union(self, bases, offs, tau, threads=8, self_check=False):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        po, pv = C.c_void_p(), C.c_void_p()
        if self._L.fo_threshold_union(self._h, _ptr(bases), _ptr(offs), n, float(tau), C.byref(po), C.byref(pv), threads,
                                      int(self_check)) != 0:
            raise RuntimeError("oracle: %s" % self._L.fo_last_error().decode())
        return _take(self._L, n, po, pv)

    def kmer_conservation(self, seq):
        """index::kmer_conservation -> list of (start_pos_in_query, num_kmers, color_set_id)"""
        seq = bytes(seq)
        n, p = C.c_uint64(), C.c_void_p()
        self._L.fo_kmer_conservation(self._h, seq, len(seq), C.byref(n), C.byref(p))
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(max(1, 3 * n.value),))[:3 * n.value].copy()
        self._L.fo_free(p)
        return [tuple(int(x) for x in a[3 * i:3 * i + 3]) for i in range(n.value)]

    def kmer_matches(self, seq):
        """index::kmer_matches -> (positive flags per k-mer, counts per colour)"""
        seq = bytes(seq)
        i = self.info()
        nk = max(0, len(seq) - i["k"] + 1)
        pos = np.zeros(max(1, nk), dtype=np.uint8)
        cnt = np.zeros(i["num_colors"], dtype=np.uint32)
        self._L.fo_kmer_matches(self._h, seq, len(seq), _ptr(pos), _ptr(cnt))
        return pos[:nk], cnt

    def time_pseudoalign(self, bases, offs, algo=0, tau=0.8, threads=8):
        """returns (seconds, mapped reads, total colours) for the reference-style worker loop"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        m, t = C.c_uint64(), C.c_uint64()
        sec = self._L.fo_time_pseudoalign(self._h, _ptr(bases), _ptr(offs), len(offs) - 1, algo, float(tau), threads,
                                          C.byref(m), C.byref(t))
        return sec, m.value, t.value

    def format_ascii(self, offs, colors, first_id=0):
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        colors = np.ascontiguousarray(colors, dtype=np.uint32)
        ln = C.c_uint64()
        p = self._L.fo_format_ascii(_ptr(offs), _ptr(colors), len(offs) - 1, first_id, C.byref(ln))
        s = C.string_at(p, ln.value)
        self._L.fo_free(p)
        return s

    def close(self):
        if self._h:
            self._L.fo_index_free(self._h)
            self._h = None

    def __del__(self):
        self.close()
