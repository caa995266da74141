"""CPU suite part 1: the oracle (C++ restatement of the reference algorithms) against the golden vectors
produced by the independent k-mer-level oracle, and against the reference's own brute-force definitions
(util::check_intersection / check_union restated; self_check=True)."""
import numpy as np
import pytest

from conftest import csr_to_lists, load_golden_reads, load_golden_tsv
from fulgor_amd import pack_reads


def test_oracle_full_intersection_matches_golden(s10_oracle):
    reads = load_golden_reads()
    b, o = pack_reads(reads)
    offs, cols = s10_oracle.full_intersection(b, o, threads=4, self_check=True)
    assert csr_to_lists(offs, cols) == load_golden_tsv("s10_full_intersection.tsv")


@pytest.mark.parametrize("tau", [0.8, 1.0, 0.01])
def test_oracle_threshold_union_matches_golden(s10_oracle, tau):
    reads = load_golden_reads()
    b, o = pack_reads(reads)
    offs, cols = s10_oracle.threshold_union(b, o, tau, threads=4, self_check=True)
    assert csr_to_lists(offs, cols) == load_golden_tsv("s10_threshold_union_%s.tsv" % tau)


def test_oracle_intersect_ids_equals_two_step(s10_oracle):
    reads = load_golden_reads()
    b, o = pack_reads(reads)
    ido, ids = s10_oracle.fetch_color_set_ids(b, o, threads=4)
    lists = csr_to_lists(ido, ids)
    assert all(l == sorted(set(l)) for l in lists)  # sorted unique (ps_full_intersection.cpp:371-373)
    o2, c2 = s10_oracle.intersect_ids(ids, ido, threads=4, self_check=True)
    o1, c1 = s10_oracle.full_intersection(b, o, threads=4)
    assert np.array_equal(o1, o2) and np.array_equal(c1, c2)


def test_oracle_ascii_format(s10_oracle):
    offs = np.array([0, 3, 3, 4], dtype=np.uint64)
    cols = np.array([1, 20, 300, 7], dtype=np.uint32)
    assert s10_oracle.format_ascii(offs, cols, first_id=5) == b"5\t3\t1\t20\t300\n6\t0\n7\t1\t7\n"


def test_oracle_strand_symmetry(s10_oracle):
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    reads = load_golden_reads()[:300]
    rc = [r.translate(comp)[::-1] for r in reads]
    b1, o1 = pack_reads(reads)
    b2, o2 = pack_reads(rc)
    assert csr_to_lists(*s10_oracle.full_intersection(b1, o1)) == csr_to_lists(*s10_oracle.full_intersection(b2, o2))
    assert csr_to_lists(*s10_oracle.threshold_union(b1, o1, 0.8)) == csr_to_lists(*s10_oracle.threshold_union(b2, o2, 0.8))


# ---- the other three codecs: same colour numbering and colour-set ids => same golden vectors -----------
CODEC_CASES = [(1, 3, 4), (1, 10, 1), (2, 3, 4), (2, 4, 2), (2, 1, 16), (3, 3, 4), (3, 4, 2), (3, 10, 1), (3, 1, 16)]


@pytest.mark.parametrize("index_type,psize,csize", CODEC_CASES)
def test_oracle_meta_diff_metadiff_match_golden(s10_dump, index_type, psize, csize):
    """meta / differential / meta-differential cursors + meta_intersect, diff_intersect, merge_meta,
    merge_diff, merge_metadiff (restated) against the vectors of the independent k-mer oracle"""
    from oracle.pyoracle import OracleIndex
    orc = OracleIndex.from_dump(s10_dump).convert(index_type, psize, csize)
    b, o = pack_reads(load_golden_reads())
    assert csr_to_lists(*orc.full_intersection(b, o, threads=4)) == load_golden_tsv("s10_full_intersection.tsv")
    for tau in (0.8, 1.0, 0.01):
        assert csr_to_lists(*orc.threshold_union(b, o, tau, threads=4)) == load_golden_tsv("s10_threshold_union_%s.tsv" % tau)


def _mask_of_set(orc, csid):
    o, c = orc.intersect_ids(np.array([csid], dtype=np.uint32), np.array([0, 1], dtype=np.uint64), threads=1)
    return sum(1 << int(x) for x in c)


def test_oracle_kmer_level_queries_match_golden(s10_oracle):
    """kmer_matches / kmer_conservation restatements against per-k-mer colour masks computed from the genomes"""
    from conftest import load_golden_kmer_level
    reads = load_golden_reads()
    for rid, (flags, counts, runs) in load_golden_kmer_level().items():
        pos, cnt = s10_oracle.kmer_matches(reads[rid])
        assert np.array_equal(pos, flags) and np.array_equal(cnt, counts if len(flags) else np.zeros(10, np.uint32)), rid
        got = [(s, n, _mask_of_set(s10_oracle, cs)) for s, n, cs in s10_oracle.kmer_conservation(reads[rid])]
        assert got == runs, rid


# ---- above 64 colours: the seeded 256-genome collection (tests/golden/synth_c256.py, make_golden_c256.py) --------------
@pytest.mark.parametrize("index_type,psize,csize", [(0, 0, 0), (1, 16, 4), (2, 48, 8), (3, 48, 8)])
def test_oracle_matches_golden_at_256_colours(c256_dump, index_type, psize, csize):
    """the restatement (hybrid lists of all three kinds at n = 256, and the three other codecs) against vectors that the
    independent k-mer oracle computed straight from the 256 genomes"""
    from oracle.pyoracle import OracleIndex
    orc = OracleIndex.from_dump(c256_dump)
    if index_type:
        orc.convert(index_type, psize, csize)
    b, o = pack_reads(load_golden_reads("c256_reads.fa"))
    offs, cols = orc.full_intersection(b, o, threads=4, self_check=index_type == 0)
    assert csr_to_lists(offs, cols) == load_golden_tsv("c256_full_intersection.tsv")
    for tau in (0.8, 0.3):
        offs, cols = orc.threshold_union(b, o, tau, threads=4, self_check=index_type == 0)
        assert csr_to_lists(offs, cols) == load_golden_tsv("c256_threshold_union_%s.tsv" % tau)


@pytest.mark.parametrize("index_type,psize,csize", [(0, 0, 0), (1, 4546, 16), (2, 160, 1), (3, 160, 16)])
def test_oracle_matches_golden_at_4546_colours(s4546small_dump, index_type, psize, csize):
    """golden vectors AT 4546 COLOURS, computed by oracle/dump_oracle.py straight from the dump text (colour set of a k-mer = colour
    set of its unitig; python sets and numpy, nothing of the restatement): the restatement — built from the same dump files, with
    its own encoder, cursors, `intersect` and `merge` — must reproduce them (tests/golden/make_golden_s4546small.py)"""
    from oracle.pyoracle import OracleIndex
    _, base = s4546small_dump
    orc = OracleIndex.from_dump(base)
    if index_type:
        orc.convert(index_type, psize, csize)  # the restated differential / meta / meta-differential cursors and merges
    b, o = pack_reads(load_golden_reads("s4546small_reads.fa"))
    offs, cols = orc.full_intersection(b, o, threads=8)
    assert csr_to_lists(offs, cols) == load_golden_tsv("s4546small_full_intersection.tsv.gz")
    for tau in (0.8, 0.3):
        offs, cols = orc.threshold_union(b, o, tau, threads=8)
        assert csr_to_lists(offs, cols) == load_golden_tsv("s4546small_threshold_union_%s.tsv.gz" % tau)
