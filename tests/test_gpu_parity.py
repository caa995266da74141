"""GPU suite: the HIP path, called through the C ABI, against (1) the committed golden vectors of the
independent k-mer oracle, (2) the oracle (CPU restatement of the reference) on seeded reads, bit-exact,
(3) size-independent properties at BASELINE config size (1M reads on salmonella_10)."""
import contextlib
import os
import sys

import numpy as np
import pytest

import fulgor_amd
from conftest import ROOT, S10_GENOMES, csr_to_lists, load_golden_reads, load_golden_tsv
from fulgor_amd import pack_reads

pytestmark = pytest.mark.gpu

# The colour stage has two executions (fgpu_tune, FGPU_TUNE_DENSE_ROWS): the dense rows (k2r_intersect / k3r_union, the default
# while the rows fit the device) and the packed blocks of the gap-coded lists (k2a_intersect / k3a_union: what decodes the
# reference's lists, and what a collection runs whose rows do not fit). Tests that take `colour_stage` run under both.
COLOUR_STAGES = [pytest.param(True, id="dense-rows"), pytest.param(False, id="packed-blocks")]


@pytest.fixture(params=COLOUR_STAGES)
def colour_stage(request):
    return request.param


_ORACLE_MEMO = {}


def oracle_once(key, fn):
    """the oracle's answer does not depend on how the GPU executes: computed once per test and inputs, shared by the two stages"""
    if key not in _ORACLE_MEMO:
        _ORACLE_MEMO[key] = fn()
    return _ORACLE_MEMO[key]


@contextlib.contextmanager
def stage(ix, dense_rows):
    """the index with the colour stage set to dense rows / packed blocks (the session's indexes go back to the default)"""
    ix.tune(dense_rows=dense_rows)
    try:
        yield ix
    finally:
        ix.tune(dense_rows=True)


def test_gpu_full_intersection_matches_golden(s10_gpu, colour_stage):
    b, o = pack_reads(load_golden_reads())
    with stage(s10_gpu, colour_stage):
        offs, cols = s10_gpu.pseudoalign_full_intersection_batch(b, o)
    assert csr_to_lists(offs, cols) == load_golden_tsv("s10_full_intersection.tsv")


@pytest.mark.parametrize("tau", [0.8, 1.0, 0.01])
def test_gpu_threshold_union_matches_golden(s10_gpu, tau, colour_stage):
    b, o = pack_reads(load_golden_reads())
    with stage(s10_gpu, colour_stage):
        offs, cols = s10_gpu.pseudoalign_threshold_union_batch(b, o, tau)
    assert csr_to_lists(offs, cols) == load_golden_tsv("s10_threshold_union_%s.tsv" % tau)


@pytest.fixture(scope="module")
def seeded_reads(built):
    from fulgor_amd.reads import ReadGenerator
    g = ReadGenerator(S10_GENOMES)
    return g.generate(0, 50000, 150, 7)


def test_gpu_fetch_color_set_ids_equals_oracle(s10_gpu, s10_oracle, seeded_reads):
    b, o = seeded_reads
    go, gi = s10_gpu.fetch_color_set_ids_batch(b, o)
    oo, oi = s10_oracle.fetch_color_set_ids(b, o)
    assert np.array_equal(go, oo) and np.array_equal(gi, oi)


def test_gpu_full_intersection_equals_oracle(s10_gpu, s10_oracle, seeded_reads, colour_stage):
    b, o = seeded_reads
    with stage(s10_gpu, colour_stage):
        go, gc = s10_gpu.pseudoalign_full_intersection_batch(b, o)
    oo, oc = oracle_once("s10_fi", lambda: s10_oracle.full_intersection(b, o))
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)


@pytest.mark.parametrize("tau", [0.8, 0.5, 1.0])
def test_gpu_threshold_union_equals_oracle(s10_gpu, s10_oracle, seeded_reads, tau, colour_stage):
    b, o = seeded_reads
    with stage(s10_gpu, colour_stage):
        go, gc = s10_gpu.pseudoalign_threshold_union_batch(b, o, tau)
    oo, oc = oracle_once(("s10_tu", tau), lambda: s10_oracle.threshold_union(b, o, tau))
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)


def test_gpu_intersect_ids_equals_oracle(s10_gpu, s10_oracle, seeded_reads, colour_stage):
    b, o = seeded_reads
    ido, ids = s10_oracle.fetch_color_set_ids(b, o)
    # random id lists (not produced by any read): every hybrid encoding mixed, up to 40 lists
    rng = np.random.default_rng(3)
    ns = s10_gpu.num_color_sets()
    lens = rng.integers(0, 40, size=3000)
    lists = [np.unique(rng.integers(0, ns, size=l)).astype(np.uint32) for l in lens]
    ido2 = np.zeros(len(lists) + 1, dtype=np.uint64)
    ido2[1:] = np.cumsum([len(l) for l in lists])
    ids2 = np.concatenate(lists) if lists else np.zeros(0, dtype=np.uint32)
    with stage(s10_gpu, colour_stage):
        go, gc = s10_gpu.intersect_ids_batch(ids, ido)
        go2, gc2 = s10_gpu.intersect_ids_batch(ids2, ido2)
    oo, oc = oracle_once("s10_ids", lambda: s10_oracle.intersect_ids(ids, ido))
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)
    oo, oc = oracle_once("s10_ids2", lambda: s10_oracle.intersect_ids(ids2, ido2, self_check=True))
    assert np.array_equal(go2, oo) and np.array_equal(gc2, oc)


def test_gpu_per_read_members(s10_gpu, s10_oracle):
    reads = load_golden_reads()[:20]
    gold = load_golden_tsv("s10_full_intersection.tsv")
    for i, r in enumerate(reads):
        ids = s10_gpu.fetch_color_set_ids(r)
        assert s10_gpu.pseudoalign_full_intersection(ids) == gold[i]


def test_gpu_edge_batches(s10_gpu):
    # empty batch
    offs, cols = s10_gpu.pseudoalign_full_intersection_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert offs.tolist() == [0] and len(cols) == 0
    # only empty / too-short reads
    b, o = pack_reads([b"", b"ACGT", b"A" * 30])
    offs, cols = s10_gpu.pseudoalign_full_intersection_batch(b, o)
    assert offs.tolist() == [0, 0, 0, 0]
    offs, cols = s10_gpu.pseudoalign_threshold_union_batch(b, o, 0.8)
    assert offs.tolist() == [0, 0, 0, 0]
    with pytest.raises(RuntimeError, match="threshold"):
        s10_gpu.pseudoalign_threshold_union_batch(b, o, 0.0)
    with pytest.raises(RuntimeError, match="threshold"):
        s10_gpu.pseudoalign_threshold_union_batch(b, o, 1.5)
    # offsets that are not monotone are refused, also when they step DOWN by a constant (equal unsigned differences: ADVICE r5)
    base = np.frombuffer(b"ACGT" * 100, dtype=np.uint8)
    for bad in ([0, 200, 100, 300], [0, (1 << 64) - 150, (1 << 64) - 300, (1 << 64) - 450]):
        with pytest.raises(RuntimeError, match="monotone"):
            s10_gpu.pseudoalign_full_intersection_batch(base, np.array(bad, dtype=np.uint64))


@pytest.mark.parametrize("windows", [1.5, 2, 3, 4])
def test_gpu_reads_up_to_512_kmers(s10_gpu, s10_oracle, windows, colour_stage):
    """batches whose longest read has 129..512 k-mers (159- to 542-base reads) run the lookup kernel's instantiations for units of up to
    192 (round 6), 256 and 512 k-mers: every length around the window boundaries, invalid bases in several windows, substitutions."""
    from oracle.kmer_oracle import read_fasta
    rng = np.random.default_rng(250 + int(2 * windows))
    src = max(read_fasta(S10_GENOMES[3]), key=len)
    top = int(128 * windows) + 30
    lens = [top, top - 1, top - 36, 250, 159, 160, 158, 157, 191, 192, 193, 222, 223, 224, 200, 100, 31, 30, 0, 64, 128, 129, top]
    for j in range(1, int(windows)):
        lens += [128 * j + 29, 128 * j + 30, 128 * j + 31, 128 * j + 94]
    lens = [min(l, top) for l in lens]
    reads = []
    for i, l in enumerate(lens * 6):
        st = int(rng.integers(0, len(src) - 600))
        r = bytearray(src[st:st + l])
        if i % 3 == 1 and l > 40:  # substitutions anywhere
            for p_ in rng.integers(0, l, 3):
                r[p_] = b"ACGT"[int(rng.integers(0, 4))]
        if i % 5 == 2 and l > 40:  # invalid bases: in the first window and further on
            r[int(rng.integers(0, min(l, 128)))] = ord("N")
            r[int(rng.integers(l // 2, l))] = ord("N")
            r[int(rng.integers(0, l))] = ord("N")
        if i % 7 == 3:
            r = bytearray(bytes(r).lower())
        reads.append(bytes(r))
    b, o = pack_reads(reads)
    with stage(s10_gpu, colour_stage):
        got3 = (s10_gpu.pseudoalign_full_intersection_batch(b, o), s10_gpu.pseudoalign_threshold_union_batch(b, o, 0.8),
                s10_gpu.fetch_color_set_ids_batch(b, o))
    want3 = oracle_once(("s10_512", windows), lambda: (s10_oracle.full_intersection(b, o), s10_oracle.threshold_union(b, o, 0.8), s10_oracle.fetch_color_set_ids(b, o)))
    for got, want in zip(got3, want3):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # per-k-mer ids keep their order across the two windows
    from fulgor_amd.index import conservation_triples
    ko, ki = s10_gpu.kmer_color_set_ids_batch(b, o)
    for j, r in enumerate(reads):
        ids = ki[int(ko[j]):int(ko[j + 1])]
        assert len(ids) == max(0, len(r) - 31 + 1)
        assert conservation_triples(ids) == s10_oracle.kmer_conservation(r)


@pytest.mark.parametrize("top", [144, 100, 145])
def test_gpu_batches_of_short_reads(s10_gpu, s10_oracle, top):
    """batches whose longest read has at most 114 k-mers (144 bases) run the SHORT instantiation of the lookup kernel (two rounds of
    m-mer positions instead of three; round 6); 145 bases: the general one. Every length up to the top, substitutions, invalid bases,
    lower case; ids, both algorithms and the per-k-mer ids against the oracle."""
    from oracle.kmer_oracle import read_fasta
    from fulgor_amd.index import conservation_triples
    rng = np.random.default_rng(top)
    src = max(read_fasta(S10_GENOMES[5]), key=len)
    lens = list(range(0, top + 1)) * 3 + [top] * 200 + [top - 1] * 50 + [31] * 20
    reads = []
    for i, l in enumerate(lens):
        st = int(rng.integers(0, len(src) - 300))
        r = bytearray(src[st:st + l])
        if i % 3 == 1 and l > 40:
            for p_ in rng.integers(0, l, 2):
                r[p_] = b"ACGT"[int(rng.integers(0, 4))]
        if i % 5 == 2 and l > 40:
            r[int(rng.integers(0, l))] = ord("N")
        if i % 7 == 3:
            r = bytearray(bytes(r).lower())
        reads.append(bytes(r))
    b, o = pack_reads(reads)
    got3 = (s10_gpu.pseudoalign_full_intersection_batch(b, o), s10_gpu.pseudoalign_threshold_union_batch(b, o, 0.8), s10_gpu.fetch_color_set_ids_batch(b, o))
    want3 = (s10_oracle.full_intersection(b, o), s10_oracle.threshold_union(b, o, 0.8), s10_oracle.fetch_color_set_ids(b, o))
    for got, want in zip(got3, want3):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    ko, ki = s10_gpu.kmer_color_set_ids_batch(b[:int(o[300])], o[:301])
    for j in range(300):
        assert conservation_triples(ki[int(ko[j]):int(ko[j + 1])]) == s10_oracle.kmer_conservation(reads[j])


def test_gpu_threshold_union_reads_of_128_to_255_kmers(s10_gpu, s10_oracle, colour_stage):
    """batches whose longest read has 128..255 k-mers (e.g. 250-base reads) keep 8-bit score counters, unbiased, and
    compare them with the threshold byte-wise: thresholds on both sides of 128, zero, and equal to the score; chimeric reads
    (many colour sets, complemented lists included); the scores themselves through kmer_matches."""
    from oracle.kmer_oracle import read_fasta
    rng = np.random.default_rng(255)
    srcs = [max(read_fasta(g), key=len) for g in S10_GENOMES[:4]]
    reads = []
    for i in range(160):
        l = int(rng.integers(158, 286)) if i else 285
        src = srcs[i % 4]
        st = int(rng.integers(0, len(src) - 400))
        r = bytearray(src[st:st + l])
        if i % 4 == 1:  # chimera of two genomes
            o2 = srcs[(i + 1) % 4]
            st2 = int(rng.integers(0, len(o2) - 400))
            r[l // 2:] = o2[st2:st2 + l - l // 2]
        if i % 3 == 2:
            for p_ in rng.integers(0, l, 4):
                r[p_] = b"ACGT"[int(rng.integers(0, 4))]
        if i % 9 == 4:
            r[int(rng.integers(0, l))] = ord("N")
        reads.append(bytes(r))
    reads += [srcs[0][1000:1100], srcs[1][5000:5031], b""]  # shorter reads in the same batch
    b, o = pack_reads(reads)
    for tau in (0.001, 0.3, 0.5, 0.55, 0.8, 1.0):
        with stage(s10_gpu, colour_stage):
            got = s10_gpu.pseudoalign_threshold_union_batch(b, o, tau)
        want = oracle_once(("s10_255", tau), lambda: s10_oracle.threshold_union(b, o, tau))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), tau
    mo, pos, counts = s10_gpu.kmer_matches_batch(b, o)
    for j in range(0, len(reads), 7):
        opos, ocnt = s10_oracle.kmer_matches(reads[j])
        assert np.array_equal(pos[int(mo[j]):int(mo[j + 1])], opos)
        assert np.array_equal(counts[j], ocnt)


def test_gpu_reads_of_any_length(s10_gpu, s10_oracle):
    """ragged batch: 31 bp .. 300 kbp. Reads above 512 k-mers are cut into overlapping segments by the host
    and their id lists merged on the device (k_merge_segments); answers must not depend on that."""
    from oracle.kmer_oracle import read_fasta
    src = max(read_fasta(S10_GENOMES[5]), key=len)
    lens = [1054, 700, 300, 151, 1000, 31, 64, 95, 96, 159, 1055, 2078, 5000, 60000, 30, 0, 1100]
    reads = [src[i * 1000:i * 1000 + l] for i, l in enumerate(lens)]
    reads.append(src[200000:203000].replace(b"A", b"N", 3))  # long read with invalid windows
    reads.append(src[300000:400000])  # 196 segments of 512 k-mers: several segments per lane in the device merge
    reads.append(src[50000:350000])  # 586 segments: beyond the LDS cursors of the merge kernel
    reads.append(src[300000:330000] + b"N" + src[300000:330000])  # every id list occurs in two groups of segments
    b, o = pack_reads(reads)
    for got, want in ((s10_gpu.pseudoalign_full_intersection_batch(b, o), s10_oracle.full_intersection(b, o)),
                      (s10_gpu.pseudoalign_threshold_union_batch(b, o, 0.8), s10_oracle.threshold_union(b, o, 0.8)),
                      (s10_gpu.pseudoalign_threshold_union_batch(b, o, 1.0), s10_oracle.threshold_union(b, o, 1.0)),
                      (s10_gpu.fetch_color_set_ids_batch(b, o), s10_oracle.fetch_color_set_ids(b, o))):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # the same batch through the device-resident API in two passes
    rd = s10_gpu.upload_reads(b, o)
    res = s10_gpu.new_result()
    s10_gpu.run(rd, res, fulgor_amd.FULL_INTERSECTION, first=10, count=6)
    go, gc = res.download()
    wo, wc = s10_oracle.full_intersection(b[int(o[10]):int(o[16])], o[10:17] - o[10])
    assert np.array_equal(go, wo) and np.array_equal(gc, wc)


def test_gpu_config_size_1M_reads_bit_exact_and_properties(s10_gpu, s10_oracle, built):
    """BASELINE configs[1]: salmonella_10, 1M synthetic 150 bp reads, full-intersection, bit-exact vs CPU."""
    from fulgor_amd.reads import ReadGenerator
    g = ReadGenerator(S10_GENOMES)
    b, o = g.generate(0, 1_000_000, 150, 42)
    reads = s10_gpu.upload_reads(b, o)
    res = s10_gpu.new_result()
    s10_gpu.run(reads, res, fulgor_amd.FULL_INTERSECTION)
    n, total, mapped = res.sizes()
    go, gc = res.download()
    oo, oc = s10_oracle.full_intersection(b, o)
    assert n == 1_000_000 and np.array_equal(go, oo) and np.array_equal(gc, oc)
    # properties: strictly increasing per read, in range, mapped counter, idempotence of a second pass
    d = np.diff(gc.astype(np.int64))
    starts = go[1:-1].astype(np.int64)
    starts = starts[(starts > 0) & (starts < len(gc))]
    d[starts - 1] = 1
    assert (d > 0).all() and gc.max() < s10_gpu.num_colors()
    assert mapped == int((np.diff(go.astype(np.int64)) > 0).sum())
    s10_gpu.run(reads, res, fulgor_amd.FULL_INTERSECTION)
    go2, gc2 = res.download()
    assert np.array_equal(go, go2) and np.array_equal(gc, gc2)
    # chunked passes give the same answers as one pass
    s10_gpu.run(reads, res, fulgor_amd.FULL_INTERSECTION, first=300_000, count=200_000)
    go3, gc3 = res.download()
    a, bnd = int(go[300_000]), int(go[500_000])
    assert np.array_equal(gc3, gc[a:bnd]) and np.array_equal(go3.astype(np.int64), go[300_000:500_001].astype(np.int64) - a)
    # threshold-union at the same size
    s10_gpu.run(reads, res, fulgor_amd.THRESHOLD_UNION, 0.8)
    to, tc = res.download()
    oo, oc = s10_oracle.threshold_union(b, o, 0.8)
    assert np.array_equal(to, oo) and np.array_equal(tc, oc)


def test_gpu_hit_counts(s10_gpu, seeded_reads):
    import torch
    b, o = seeded_reads
    reads = s10_gpu.upload_reads(b, o)
    res = s10_gpu.new_result()
    s10_gpu.run(reads, res, fulgor_amd.FULL_INTERSECTION)
    n = s10_gpu.num_colors()
    hits = torch.zeros(n + 2, dtype=torch.int64, device="cuda:0")
    res.accumulate_hits(hits.data_ptr())
    res.accumulate_hits(hits.data_ptr())
    go, gc = res.download()
    want = np.bincount(gc, minlength=n)
    got = hits.cpu().numpy()
    assert np.array_equal(got[:n], 2 * want)
    assert got[n] == 2 * 50000 and got[n + 1] == 2 * int((np.diff(go.astype(np.int64)) > 0).sum())


@pytest.mark.parametrize("algo,tau", [(fulgor_amd.FULL_INTERSECTION, 0.0), (fulgor_amd.THRESHOLD_UNION, 0.8)])
def test_colour_lists_are_materialised_only_on_demand(s4546small, colour_stage, algo, tau):
    """round-4 review, item 2: a pass leaves rows / small-result slots / sizes; the compressed records, the two counters and the
    per-colour hit vector come from those and the expansion kernel never runs; download (or expand) runs it once, and the hit
    vector counted from the rows equals the one the expansion kernel's histogram gives. 4546 colours: results of at most 16
    colours travel as colours, not as rows (both paths of the formatter and of the hit counter)."""
    import torch
    from fulgor_amd.driver import Formatter, hit_vector
    from oracle.pyoracle import parse_compressed
    ix, _, gen, _, _ = s4546small
    b, o = gen.generate(500, 30000, 150, 9)
    n, nr = ix.num_colors(), len(o) - 1
    with stage(ix, colour_stage):
        rd, res, res2 = ix.upload_reads(b, o), ix.new_result(), ix.new_result()
        ix.timing_enable(True)
        ix.timing_reset()
        ix.run(rd, res, algo, tau)
        _, total, mapped = res.sizes()
        rec = bytes(res.format_view(2, 11))
        lazy = torch.zeros(n + 2, dtype=torch.int64, device="cuda:0")
        res.accumulate_hits(lazy.data_ptr())
        assert ix.timing()["k2b_expand"][1] == 0, "the expansion kernel ran although nobody asked for the colour lists"
        go, gc = res.download()
        assert ix.timing()["k2b_expand"][1] == 1
        go2, gc2 = res.download()  # (materialised once)
        assert ix.timing()["k2b_expand"][1] == 1 and np.array_equal(gc, gc2)
        ix.run(rd, res2, algo, tau)
        res2.expand()
        folded = torch.zeros(n + 2, dtype=torch.int64, device="cuda:0")
        res2.accumulate_hits(folded.data_ptr())
        from_lists, from_rows = res2.checksum()
        ix.timing_enable(False)
    # the two device-side checksums of the colour lists (what the config-size tests rely on) against numpy on the downloaded lists
    v = (gc.astype(np.uint64) + np.uint64(1)) * (np.arange(len(gc), dtype=np.uint64) + np.uint64(1))
    with np.errstate(over="ignore"):
        want_sum = (len(gc), int(v.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(v * np.uint64(0x9E3779B97F4A7C15))) if len(gc) else 0)
    assert tuple(int(x) for x in from_lists) == want_sum and tuple(int(x) for x in from_rows) == want_sum
    assert total == len(gc) and mapped == int((np.diff(go.astype(np.int64)) > 0).sum())
    ids, po, pc = parse_compressed(Formatter("compressed", n).header + rec)
    assert np.array_equal(ids, np.arange(11, 11 + nr, dtype=np.uint32)) and np.array_equal(po, go) and np.array_equal(pc, gc)
    want = hit_vector(go, gc, n)
    assert np.array_equal(lazy.cpu().numpy(), want) and np.array_equal(folded.cpu().numpy(), want)
    sizes = np.diff(go.astype(np.int64))
    if algo == fulgor_amd.FULL_INTERSECTION:
        assert ((sizes > 0) & (sizes <= 16)).sum() > 100 and (sizes > 16).sum() > 100  # both kinds of result occur


# ---- synthetic salmonella_4546-shaped index (n = 4546: sparse, bitmap and complement lists of real size) ----
@pytest.fixture(scope="module")
def s4546(built):
    import os
    from conftest import DATA
    from fulgor_amd import synth
    from fulgor_amd.reads import ReadGenerator
    from oracle.pyoracle import OracleIndex
    fg, extra = synth.ensure_s4546(DATA, S10_GENOMES)
    ix = fulgor_amd.Index(fg, device=0)
    orc = OracleIndex.from_export(ix.export())  # raises if a k-mer occurs in two unitigs
    gen = ReadGenerator(S10_GENOMES, raw_sequences=extra)
    return ix, orc, gen


def test_s4546_full_intersection_equals_oracle(s4546, colour_stage):
    ix, orc, gen = s4546
    b, o = gen.generate(0, 30000, 150, 42)
    with stage(ix, colour_stage):
        go, gc = ix.pseudoalign_full_intersection_batch(b, o)
    oo, oc = oracle_once("s4546_fi", lambda: orc.full_intersection(b, o, threads=32))
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)
    if colour_stage:
        orc.full_intersection(b[:150 * 500], o[:501], threads=8, self_check=True)  # restatement vs check_intersection
    i1, d1 = ix.fetch_color_set_ids_batch(b, o)
    i2, d2 = oracle_once("s4546_ids", lambda: orc.fetch_color_set_ids(b, o, threads=32))
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)


@pytest.mark.parametrize("tau", [0.8, 0.3, 1.0, 0.5])
def test_s4546_threshold_union_equals_oracle(s4546, tau, colour_stage):
    """(k3r_union: tau 0.8 and 1.0 = five planes of deficit counters for the reads the multiplexer tree does not take, 0.5 = six,
    0.3 = byte counters; seven planes: test_s4546_longer_reads_equal_oracle at 500 bases)"""
    ix, orc, gen = s4546
    b, o = gen.generate(100000, 20000, 150, 42)
    with stage(ix, colour_stage):
        go, gc = ix.pseudoalign_threshold_union_batch(b, o, tau)
    oo, oc = oracle_once(("s4546_tu", tau), lambda: orc.threshold_union(b, o, tau, threads=32))
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)
    if colour_stage:
        orc.threshold_union(b[:150 * 300], o[:301], tau, threads=8, self_check=True)


@pytest.mark.parametrize("read_len", [250, 300, 500])
def test_s4546_longer_reads_equal_oracle(s4546, read_len, colour_stage):
    """4546 colours, reads of 220 / 270 / 470 k-mers: the windowed lookup kernel (2, 3, 4 windows) and the threshold union's
    plain 8-bit (250) and 16-bit (300, 500) counters against the oracle"""
    ix, orc, gen = s4546
    b, o = gen.generate(300000, 6000, read_len, 42)
    with stage(ix, colour_stage):
        go, gc = ix.pseudoalign_full_intersection_batch(b, o)
    oo, oc = oracle_once(("s4546_long_fi", read_len), lambda: orc.full_intersection(b, o, threads=32))
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)
    for tau in (0.8, 0.3):
        with stage(ix, colour_stage):
            go, gc = ix.pseudoalign_threshold_union_batch(b, o, tau)
        oo, oc = oracle_once(("s4546_long_tu", read_len, tau), lambda: orc.threshold_union(b, o, tau, threads=32))
        assert np.array_equal(go, oo) and np.array_equal(gc, oc), tau


def test_s4546_random_id_lists(s4546, colour_stage):
    """intersections of arbitrary colour-set ids: many sparse lists, > 64 lists per read, all encodings"""
    ix, orc, _ = s4546
    rng = np.random.default_rng(11)
    ns = ix.num_color_sets()
    lens = np.concatenate([rng.integers(0, 12, size=1500), rng.integers(60, 140, size=40)])
    lists = [np.unique(rng.integers(0, ns, size=l)).astype(np.uint32) for l in lens]
    ido = np.zeros(len(lists) + 1, dtype=np.uint64)
    ido[1:] = np.cumsum([len(l) for l in lists])
    ids = np.concatenate(lists)
    with stage(ix, colour_stage):
        go, gc = ix.intersect_ids_batch(ids, ido)
    oo, oc = oracle_once("s4546_random_ids", lambda: orc.intersect_ids(ids, ido, threads=32, self_check=True))
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)


def test_cli_output_is_byte_identical_to_reference_format(s10_gpu, s10_fgidx, tmp_path):
    """`fulgor pseudoalign -i .. -q .. -o ..` drop-in: ascii output equals the golden file byte for byte
    (single worker => file order), binary output decodes to the same lists"""
    import os
    from conftest import GOLDEN
    from fulgor_amd import cli
    q = os.path.join(GOLDEN, "s10_reads.fa")
    out = tmp_path / "out.tsv"
    assert cli.main(["pseudoalign", "-i", s10_fgidx, "-q", q, "-o", str(out), "--verbose"]) == 0
    assert out.read_bytes() == open(os.path.join(GOLDEN, "s10_full_intersection.tsv"), "rb").read()
    out2 = tmp_path / "out_tu.tsv"
    assert cli.main(["pseudoalign", "-i", s10_fgidx, "-q", q, "-o", str(out2), "-r", "0.8"]) == 0
    assert out2.read_bytes() == open(os.path.join(GOLDEN, "s10_threshold_union_0.8.tsv"), "rb").read()
    out3 = tmp_path / "out.bin"
    assert cli.main(["pseudoalign", "-i", s10_fgidx, "-q", q, "-o", str(out3), "--format", "binary"]) == 0
    raw = np.frombuffer(out3.read_bytes(), dtype="<u4")
    gold = load_golden_tsv("s10_full_intersection.tsv")
    p = 0
    for i, cols in enumerate(gold):
        assert raw[p] == i and raw[p + 1] == len(cols) and raw[p + 2:p + 2 + len(cols)].tolist() == cols
        p += 2 + len(cols)
    assert p == len(raw)
    # the same query file gzipped, as an ordinary stream and in blocks (bgzip): same output
    import gzip, struct, zlib
    raw = open(q, "rb").read()
    qg, qb = tmp_path / "q.fa.gz", tmp_path / "qb.fa.gz"
    qg.write_bytes(gzip.compress(raw, 6))
    with open(qb, "wb") as f:
        for at in list(range(0, len(raw), 30000)) + [len(raw)]:
            blk = raw[at:at + 30000]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            cd = co.compress(blk) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd +
                    struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk)))
    for qq in (qg, qb):
        oz = tmp_path / (qq.name + ".tsv")
        assert cli.main(["pseudoalign", "-i", s10_fgidx, "-q", str(qq), "-o", str(oz)]) == 0
        assert oz.read_bytes() == out.read_bytes()
    out5 = tmp_path / "out_dedup.tsv"
    assert cli.main(["pseudoalign", "-i", s10_fgidx, "-q", q, "-o", str(out5), "--deduplicate"]) == 0
    assert out5.read_bytes() == out.read_bytes()
    out4 = tmp_path / "out.cmp"
    assert cli.main(["pseudoalign", "-i", s10_fgidx, "-q", q, "-o", str(out4), "--format", "compressed"]) == 0
    from oracle import pyoracle
    ids, po, pc = pyoracle.parse_compressed(out4.read_bytes())
    assert ids.tolist() == list(range(len(gold))) and csr_to_lists(po, pc) == gold


# ---- meta / differential / meta-differential codecs ---------------------
CODECS = [(fulgor_amd.DIFF, 10, 4), (fulgor_amd.DIFF, 10, 1), (fulgor_amd.META, 3, 1), (fulgor_amd.META, 4, 1),
          (fulgor_amd.META_DIFF, 3, 4), (fulgor_amd.META_DIFF, 4, 2), (fulgor_amd.META_DIFF, 1, 16)]


@pytest.mark.parametrize("index_type,psize,csize", CODECS)
def test_gpu_codecs_match_golden_and_oracle(s10_fgidx, s10_dump, seeded_reads, index_type, psize, csize):
    from oracle.pyoracle import OracleIndex
    ix = fulgor_amd.Index(s10_fgidx, device=0).convert(index_type, psize, csize)
    ix.tune(dense_rows=False)  # the codec's own kernels (k_generic), not the dense rows
    b, o = pack_reads(load_golden_reads())
    assert csr_to_lists(*ix.pseudoalign_full_intersection_batch(b, o)) == load_golden_tsv("s10_full_intersection.tsv")
    for tau in (0.8, 1.0, 0.01):
        assert csr_to_lists(*ix.pseudoalign_threshold_union_batch(b, o, tau)) == load_golden_tsv("s10_threshold_union_%s.tsv" % tau)
    # the oracle's own restatement of that codec (cursors + meta_intersect / diff_intersect / merge_*)
    orc = OracleIndex.from_dump(s10_dump).convert(index_type, psize, csize)
    b, o = seeded_reads
    b, o = b[:150 * 20000], o[:20001]
    for got, want in ((ix.pseudoalign_full_intersection_batch(b, o), orc.full_intersection(b, o)),
                      (ix.pseudoalign_threshold_union_batch(b, o, 0.8), orc.threshold_union(b, o, 0.8))):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    rng = np.random.default_rng(5)
    ns = ix.num_color_sets()
    lists = [np.unique(rng.integers(0, ns, size=l)).astype(np.uint32) for l in rng.integers(0, 30, size=1500)]
    ido = np.zeros(len(lists) + 1, dtype=np.uint64)
    ido[1:] = np.cumsum([len(l) for l in lists])
    ids = np.concatenate(lists)
    got, want = ix.intersect_ids_batch(ids, ido), orc.intersect_ids(ids, ido)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("index_type,psize,csize", [(fulgor_amd.DIFF, 4546, 16), (fulgor_amd.META, 160, 1), (fulgor_amd.META_DIFF, 160, 16)])
def test_s4546_codecs_equal_hybrid(s4546, index_type, psize, csize):
    """4546 colours: the same colour sets under another codec must give the same answers as the hybrid
    index (which is itself checked against the oracle above)"""
    from conftest import DATA
    from fulgor_amd import synth
    ix, _, gen = s4546
    b, o = gen.generate(200000, 30000, 150, 42)
    want_fi = ix.pseudoalign_full_intersection_batch(b, o)
    want_tu = ix.pseudoalign_threshold_union_batch(b, o, 0.8)
    fg, _ = synth.ensure_s4546(DATA, S10_GENOMES)
    iy = fulgor_amd.Index(fg, device=0).convert(index_type, psize, csize)
    iy.tune(dense_rows=False)  # the codec's own kernels (k_generic), not the dense rows
    got_fi = iy.pseudoalign_full_intersection_batch(b, o)
    got_tu = iy.pseudoalign_threshold_union_batch(b, o, 0.8)
    assert np.array_equal(got_fi[0], want_fi[0]) and np.array_equal(got_fi[1], want_fi[1])
    assert np.array_equal(got_tu[0], want_tu[0]) and np.array_equal(got_tu[1], want_tu[1])
    # 250-base reads: both engines keep plain 8-bit score counters (128..255 k-mers)
    b2, o2 = gen.generate(900000, 3000, 250, 42)
    for tau in (0.8, 0.3, 0.001):
        w, g_ = ix.pseudoalign_threshold_union_batch(b2, o2, tau), iy.pseudoalign_threshold_union_batch(b2, o2, tau)
        assert np.array_equal(g_[0], w[0]) and np.array_equal(g_[1], w[1]), tau
    bs, os_ = b2[:int(o2[40])], o2[:41]
    for x, y in zip(ix.kmer_matches_batch(bs, os_), iy.kmer_matches_batch(bs, os_)):
        assert np.array_equal(x, y)


def test_deduplicated_path_equals_direct_path(s4546):
    from fulgor_amd import driver
    ix, _, gen = s4546
    b, o = gen.generate(500000, 40000, 150, 42)
    want = ix.pseudoalign_full_intersection_batch(b, o)
    got = driver.deduplicated_full_intersection(ix, b, o)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_device_side_deduplication_equals_direct_path(s4546, colour_stage, tmp_path):
    """`--deduplicate` on the device (fgpu_tune(FGPU_TUNE_DEDUPLICATE); tools/pseudoalign.cpp:91-226): the reads of a pass are ordered
    by a hash of their id lists, neighbours compared exactly, the intersection runs once per group and every read takes its group's
    result. A batch in which every read occurs several times (and some not at all mapped): the same CSR as the direct path, on the
    dense rows and on the packed blocks; the number of distinct lists is at most the number of distinct reads; the command line with
    --deduplicate writes the bytes the command line without it writes, in every format."""
    import subprocess
    from conftest import DATA
    from fulgor_amd import synth
    ix, _, gen = s4546
    b, o = gen.generate(700000, 6000, 150, 42)
    rng = np.random.default_rng(3)
    pick = rng.integers(0, 6000, size=40000)  # every read about seven times, in random order
    bb = np.ascontiguousarray(np.asarray(b).reshape(6000, 150)[pick].reshape(-1))
    oo = np.arange(40001, dtype=np.uint64) * np.uint64(150)
    with stage(ix, colour_stage):
        want = ix.pseudoalign_full_intersection_batch(bb, oo)
        ix.tune(deduplicate=True)
        try:
            got = ix.pseudoalign_full_intersection_batch(bb, oo)
            rd, res = ix.upload_reads(bb, oo), ix.new_result()
            ix.run(rd, res, fulgor_amd.FULL_INTERSECTION, 0.0)
            groups = res.distinct_lists()
            tu = ix.pseudoalign_threshold_union_batch(bb[:150 * 2000], oo[:2001], 0.8)  # (the knob leaves the union alone)
        finally:
            ix.tune(deduplicate=False)
        tu_want = ix.pseudoalign_threshold_union_batch(bb[:150 * 2000], oo[:2001], 0.8)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(tu[0], tu_want[0]) and np.array_equal(tu[1], tu_want[1])
    assert 0 < groups <= len(np.unique(pick)) + 1
    if colour_stage:  # the command line, once
        fg, _ = synth.ensure_s4546(DATA, S10_GENOMES)
        q = tmp_path / "dup.fq"
        with open(q, "wb") as f:
            for i in range(12000):
                f.write(b"@r%d\n%s\n+\n%s\n" % (i, bytes(bb[150 * i:150 * (i + 1)]), b"I" * 150))
        for fmt in ("ascii", "compressed"):
            outs = []
            for extra in ([], ["--deduplicate"]):
                out = tmp_path / ("o_%s_%d" % (fmt, len(extra)))
                r = subprocess.run([sys.executable, "-m", "fulgor_amd", "pseudoalign", "-i", fg, "-q", str(q), "-o", str(out), "--format", fmt] + extra,
                                   cwd=ROOT, capture_output=True, text=True, timeout=600)
                assert r.returncode == 0, r.stdout + r.stderr
                outs.append(open(out, "rb").read())
            assert outs[0] == outs[1] and len(outs[0]) > 12000


def test_gpu_kmer_conservation_and_matches(s10_gpu, s10_oracle):
    """the reference's two other query tools on the same lookup kernel: per-k-mer colour-set ids ->
    kmer_conservation triples (src/kmer_conservation.cpp:7-54); un-thresholded union scores -> kmer_matches
    counts (src/kmer_matches.cpp:7-30). Checked against the oracle restatement and, for the counts, against
    the independent k-mer masks of the golden generator's oracle."""
    from fulgor_amd.index import conservation_triples
    reads = load_golden_reads()
    sel = list(range(0, 60)) + list(range(1000, 1010))
    sub = [reads[i] for i in sel]
    b, o = pack_reads(sub)
    ko, ki = s10_gpu.kmer_color_set_ids_batch(b, o)
    mo, pos, counts = s10_gpu.kmer_matches_batch(b, o)
    assert np.array_equal(ko, mo)
    for j, r in enumerate(sub):
        ids = ki[int(ko[j]):int(ko[j + 1])]
        assert len(ids) == max(0, len(r) - 31 + 1)
        assert conservation_triples(ids) == s10_oracle.kmer_conservation(r)
        opos, ocnt = s10_oracle.kmer_matches(r)
        assert np.array_equal(pos[int(ko[j]):int(ko[j + 1])], opos)
        assert np.array_equal(counts[j], ocnt)
    assert s10_gpu.kmer_conservation(sub[3]) == s10_oracle.kmer_conservation(sub[3])
    # independent golden vectors (per-k-mer colour masks straight from the genomes)
    from conftest import load_golden_kmer_level
    gold = load_golden_kmer_level()
    for j, rid in enumerate(sel):
        if rid not in gold:
            continue
        flags, gcounts, runs = gold[rid]
        assert np.array_equal(pos[int(ko[j]):int(ko[j + 1])], flags)
        assert np.array_equal(counts[j], gcounts)
        tr = conservation_triples(ki[int(ko[j]):int(ko[j + 1])])
        assert [(s_, n_) for s_, n_, _ in tr] == [(s_, n_) for s_, n_, _ in runs]
        for (_, _, cs), (_, _, mask) in zip(tr, runs):
            assert sum(1 << c for c in s10_gpu.pseudoalign_full_intersection([cs])) == mask
    # a long read (segmented lookup) keeps per-k-mer order
    from oracle.kmer_oracle import read_fasta
    src = max(read_fasta(S10_GENOMES[2]), key=len)[50000:53000]
    assert s10_gpu.kmer_conservation(src) == s10_oracle.kmer_conservation(src)


@pytest.mark.parametrize("index_type,psize,csize", [(fulgor_amd.DIFF, 10, 4), (fulgor_amd.META, 4, 1), (fulgor_amd.META_DIFF, 4, 4)])
def test_gpu_kmer_matches_on_other_codecs(s10_gpu, s10_fgidx, index_type, psize, csize):
    """index::kmer_matches is codec independent: the counts of the re-encoded index equal the hybrid index's"""
    reads = load_golden_reads()
    b, o = pack_reads(reads[:80] + reads[1000:1010])
    mo, pos, counts = s10_gpu.kmer_matches_batch(b, o)
    ix = fulgor_amd.Index(s10_fgidx, device=0)
    ix.convert(index_type, psize, csize)
    ix.tune(dense_rows=False)  # the codec's own kernels (k_generic), not the dense rows
    mo2, pos2, counts2 = ix.kmer_matches_batch(b, o)
    assert np.array_equal(mo, mo2) and np.array_equal(pos, pos2) and np.array_equal(counts, counts2)


def _fuzz_reads(gen, rng, n):
    """ragged, dirty reads: lengths 0..420, N runs, lower case, homopolymers, low-complexity repeats"""
    b, o = gen.generate(int(rng.integers(0, 1 << 30)), n, 150, int(rng.integers(1, 1000)))
    base = [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(n)]
    out = []
    for i, r in enumerate(base):
        kind = rng.integers(0, 10)
        if kind == 0:
            r = r[:int(rng.integers(0, 60))]
        elif kind == 1:
            r = r + base[(i + 1) % n] + base[(i + 2) % n][:int(rng.integers(0, 120))]
        elif kind == 2:
            p = int(rng.integers(0, len(r)))
            r = r[:p] + b"N" * int(rng.integers(1, 40)) + r[p:]
        elif kind == 3:
            r = r.lower()
        elif kind == 4:
            r = bytes([b"ACGT"[int(rng.integers(0, 4))]]) * int(rng.integers(1, 200))
        elif kind == 5:
            unit = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 7))).astype(np.uint8))
            r = (unit * 80)[:int(rng.integers(31, 300))]
        elif kind == 6:
            r = r[:75] + bytes([rng.choice(list(b"RYKMSWn-*"))]) + r[76:]
        out.append(r)
    return out


@pytest.mark.parametrize("which", ["s10", "s4546"])
def test_fuzz_dirty_ragged_reads(which, s10_gpu, s10_oracle, s4546, built, colour_stage):
    from fulgor_amd.reads import ReadGenerator
    rng = np.random.default_rng(2026)
    if which == "s10":
        ix, orc, gen = s10_gpu, s10_oracle, ReadGenerator(S10_GENOMES)
    else:
        ix, orc, gen = s4546
    for rep in range(3):
        reads = _fuzz_reads(gen, rng, 6000)
        b, o = pack_reads(reads)
        tau = float(rng.choice([0.25, 0.3, 0.8, 1.0]))
        with stage(ix, colour_stage):
            got3 = (ix.fetch_color_set_ids_batch(b, o), ix.pseudoalign_full_intersection_batch(b, o), ix.pseudoalign_threshold_union_batch(b, o, tau))
        want3 = oracle_once(("fuzz", which, rep), lambda: (orc.fetch_color_set_ids(b, o, threads=32), orc.full_intersection(b, o, threads=32),
                                                          orc.threshold_union(b, o, tau, threads=32)))
        for got, want in zip(got3, want3):
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def _write_wide_dump(base, rng, n=70001, k=31):
    """a small dump (the reference's text interchange format) with more than 65535 colours whose colour sets
    exercise every device form: wide offset blocks (spans > 2^16), bitmap chunks, bitmap lists, complement
    lists with few / many / zero missing colours, first and last colour alone"""
    allc = np.arange(n, dtype=np.int64)
    def pick(m):
        return np.sort(rng.choice(n, size=m, replace=False))
    q = n // 4  # (the sparse / bitmap boundary of the hybrid encoding: 17500 at 70001 colours)
    sets = [pick(50), pick(n // 14), pick(300), pick(3 * n // 7), np.setdiff1d(allc, pick(40)), allc, np.array([n - 1]),
            np.array([0]), np.setdiff1d(allc, pick(n // 7)), pick(q - 1), pick(q), np.setdiff1d(allc, pick(q)),
            np.concatenate([np.arange(100, 400), pick(64)]), np.arange(n - n // 14, n)]
    sets = [np.unique(s) for s in sets]
    unitigs = []
    for sid in range(len(sets)):
        for _ in range(3):
            unitigs.append((sid, "".join("ACGT"[c] for c in rng.integers(0, 4, size=200))))
    with open(base + ".metadata.txt", "w") as f:
        f.write("k=%d\nnum_kmers=%d\nnum_colors=%d\nnum_unitigs=%d\nnum_color_sets=%d\n" %
                (k, len(unitigs) * (200 - k + 1), n, len(unitigs), len(sets)))
    with open(base + ".filenames.txt", "w") as f:
        f.write("".join("g%d.fa\n" % i for i in range(n)))
    with open(base + ".unitigs.fa", "w") as f:
        for sid, seq in sorted(unitigs, key=lambda u: u[0]):
            f.write("> color_set_id=%d\n%s\n" % (sid, seq))
    with open(base + ".color_sets.txt", "w") as f:
        for s in sets:
            f.write("size=%d %s\n" % (len(s), " ".join(map(str, s))))
    return [u[1] for u in unitigs], len(sets)


def test_gpu_more_than_65535_colours(built, tmp_path):
    """maximum-size edge: 70001 colours (block fields wider than 16 bits, result rows of 2188 words)"""
    from oracle.pyoracle import OracleIndex
    rng = np.random.default_rng(5)
    base = str(tmp_path / "wide")
    unitigs, nsets = _write_wide_dump(base, rng)
    ix = fulgor_amd.Index(base, device=0)
    orc = OracleIndex.from_dump(base)
    assert ix.num_colors() == 70001 and ix.num_color_sets() == nsets
    reads = []
    for _ in range(400):  # chimeras of 2-3 pieces: every read meets up to 3 colour sets
        reads.append("".join(u[s:s + 60] for u, s in ((unitigs[rng.integers(len(unitigs))], rng.integers(0, 140))
                                                    for _ in range(rng.integers(1, 4)))))
    reads += [u[:150] for u in unitigs]
    b, o = pack_reads(reads)
    oo, oc = orc.full_intersection(b, o, threads=8, self_check=True)
    lists = [np.unique(rng.integers(0, nsets, size=l)).astype(np.uint32) for l in rng.integers(1, 12, size=300)]
    ido = np.zeros(len(lists) + 1, dtype=np.uint64)
    ido[1:] = np.cumsum([len(l) for l in lists])
    io, ic = orc.intersect_ids(np.concatenate(lists), ido, threads=8)
    # on the dense rows (round 5: rows of more than 32768 colours go through k2r_intersect in tiles of 32768) and on the packed blocks
    for rows in (True, False):
        ix.tune(dense_rows=rows)
        go, gc = ix.pseudoalign_full_intersection_batch(b, o)
        assert np.array_equal(go, oo) and np.array_equal(gc, oc), rows
        for tau in (0.3, 1.0):
            go, gc = ix.pseudoalign_threshold_union_batch(b, o, tau)
            uo, uc = orc.threshold_union(b, o, tau, threads=8)
            assert np.array_equal(go, uo) and np.array_equal(gc, uc), (rows, tau)
        go, gc = ix.intersect_ids_batch(np.concatenate(lists), ido)
        assert np.array_equal(go, io) and np.array_equal(gc, ic), rows
    ix.tune(dense_rows=True)
    # per-colour hit counts: this many colours do not fit the expand kernel's LDS histogram (bitmap path)
    import torch
    rd, res = ix.upload_reads(b, o), ix.new_result()
    ix.run(rd, res, fulgor_amd.FULL_INTERSECTION)
    hits = torch.zeros(70001 + 2, dtype=torch.int64, device="cuda:0")
    res.accumulate_hits(hits.data_ptr())
    go, gc = res.download()
    got = hits.cpu().numpy()
    assert np.array_equal(got[:70001], np.bincount(gc, minlength=70001)) and got[70001] == len(reads)
    # the three device formatters on wide rows (records of up to 70001 bitmap bits / tens of thousands of codes)
    from fulgor_amd.driver import Formatter
    from oracle.pyoracle import parse_compressed
    for fmt, code in (("ascii", 0), ("binary", 1)):
        assert bytes(res.format_view(code, 7)) == Formatter(fmt, 70001).add(7, go, gc)
    ids, po, pc = parse_compressed(Formatter("compressed", 70001).header + bytes(res.format_view(2, 7)))
    assert np.array_equal(ids, np.arange(7, 7 + len(reads), dtype=np.uint32)) and np.array_equal(po, go) and np.array_equal(pc, gc)


@pytest.mark.parametrize("n", [12001, 24001])
def test_gpu_dense_rows_of_two_and_four_groups_per_lane(built, tmp_path, n):
    """8193 to 32768 colours: a dense row is two resp. four 128-bit groups per lane (k2r_intersect<2>, <4>; k3r_union runs more
    rounds); intersection, union, hit counts and the compressed formatter against the oracle, and against the packed-block
    kernels of the same index"""
    import torch
    from oracle.pyoracle import OracleIndex, parse_compressed
    from fulgor_amd.driver import Formatter
    rng = np.random.default_rng(n)
    base = str(tmp_path / "mid")
    unitigs, nsets = _write_wide_dump(base, rng, n=n)
    ix = fulgor_amd.Index(base, device=0)
    orc = OracleIndex.from_dump(base)
    assert ix.num_colors() == n
    reads = []
    for _ in range(600):
        reads.append("".join(u[s:s + 60] for u, s in ((unitigs[rng.integers(len(unitigs))], rng.integers(0, 140))
                                                    for _ in range(rng.integers(1, 4)))))
    reads += [u[:150] for u in unitigs]
    b, o = pack_reads(reads)
    oo, oc = orc.full_intersection(b, o, threads=8, self_check=True)
    for rows in (True, False):
        ix.tune(dense_rows=rows)
        go, gc = ix.pseudoalign_full_intersection_batch(b, o)
        assert np.array_equal(go, oo) and np.array_equal(gc, oc), rows
        for tau in (0.3, 1.0):
            go, gc = ix.pseudoalign_threshold_union_batch(b, o, tau)
            uo, uc = orc.threshold_union(b, o, tau, threads=8)
            assert np.array_equal(go, uo) and np.array_equal(gc, uc), (rows, tau)
    ix.tune(dense_rows=True)
    lists = [np.unique(rng.integers(0, nsets, size=l)).astype(np.uint32) for l in rng.integers(1, 12, size=300)]
    ido = np.zeros(len(lists) + 1, dtype=np.uint64)
    ido[1:] = np.cumsum([len(l) for l in lists])
    go, gc = ix.intersect_ids_batch(np.concatenate(lists), ido)
    io, ic = orc.intersect_ids(np.concatenate(lists), ido, threads=8)
    assert np.array_equal(go, io) and np.array_equal(gc, ic)
    rd, res = ix.upload_reads(b, o), ix.new_result()
    ix.run(rd, res, fulgor_amd.FULL_INTERSECTION)
    hits = torch.zeros(n + 2, dtype=torch.int64, device="cuda:0")
    res.accumulate_hits(hits.data_ptr())
    go, gc = res.download()
    got = hits.cpu().numpy()
    assert np.array_equal(got[:n], np.bincount(gc, minlength=n)) and got[n] == len(reads)
    ids, po, pc = parse_compressed(Formatter("compressed", n).header + bytes(res.format_view(2, 3)))
    assert np.array_equal(ids, np.arange(3, 3 + len(reads), dtype=np.uint32)) and np.array_equal(po, go) and np.array_equal(pc, gc)
    assert bytes(res.format_view(0, 3)) == Formatter("ascii", n).add(3, go, gc)


def test_preprocessed_query_file_path_equals_direct_path(s4546, tmp_path):
    """--deduplicate through the reference's temp-file layouts: stage 1 ids -> sort/collapse -> stage 2 intersections"""
    from fulgor_amd import driver
    ix, orc, gen = s4546
    b, o = gen.generate(7000, 3000, 150, 42)
    ido, ids = ix.fetch_color_set_ids_batch(b, o)
    p1, p2 = str(tmp_path / "fetch.tmp"), str(tmp_path / "dedup.tmp")
    driver.write_fetched_ids(p1, ido, ids, first_read_id=100)
    unmapped, recs = driver.deduplicate_fetched(*driver.read_fetched_ids(p1))
    assert any(l is None for _, l in recs)  # real duplicates occur
    driver.write_preprocessed(p2, recs)
    got = {rid: cols.tolist() for rid, cols in driver.intersect_preprocessed(ix, p2, batch=700)}
    got.update({rid: [] for rid in unmapped})
    do, dc = ix.pseudoalign_full_intersection_batch(b, o)
    want = {100 + i: l for i, l in enumerate(csr_to_lists(do, dc))}
    assert got == want


@pytest.mark.parametrize("which", ["s10", "s4546"])
def test_device_formatters_are_byte_identical_to_host_formatters(which, s10_gpu, seeded_reads, s4546):
    """fgpu_result_format (HIP kernels over the resident CSR) against fgpu_formatter_add (host code, itself
    byte-identical to the restated psa_*_formatter): ascii and binary, both algorithms, first id != 0"""
    from fulgor_amd.driver import Formatter
    if which == "s10":
        ix, (b, o) = s10_gpu, seeded_reads
        b, o = b[:150 * 20000], o[:20001]
    else:
        ix, _, gen = s4546
        b, o = gen.generate(555, 6000, 150, 42)
    rd, res = ix.upload_reads(b, o), ix.new_result()
    for algo, tau in ((fulgor_amd.FULL_INTERSECTION, 0.0), (fulgor_amd.THRESHOLD_UNION, 0.6)):
        ix.run(rd, res, algo, tau)
        offs, cols = res.download()
        for fmt, code in (("ascii", 0), ("binary", 1)):
            want = Formatter(fmt, ix.num_colors()).add(4000000000 if which == "s10" else 17, offs, cols)
            got = res.format(code, 4000000000 if which == "s10" else 17)
            assert got == want
            assert bytes(res.format_view(code, 4000000000 if which == "s10" else 17)) == want  # pinned, zero-copy view
    ix.run(rd, res, fulgor_amd.FULL_INTERSECTION, 0.0, 0, 0)  # an empty pass formats to nothing
    assert res.format(0, 5) == b"" and res.format(1, 5) == b""


def test_concurrent_workers_share_one_index(s4546):
    """the reference's workers share one `const index&` (tools/pseudoalign.cpp:66-74): several host threads, each
    with its own result (own HIP stream), run different algorithms and read ranges on one index at the same time;
    every pass must equal the single-threaded answer, and chunked passes must concatenate to the one-pass result"""
    import threading
    ix, orc, gen = s4546
    b, o = gen.generate(31337, 24000, 150, 42)
    reads = ix.upload_reads(b, o)
    ref = ix.new_result()
    ix.run(reads, ref, fulgor_amd.FULL_INTERSECTION)
    fo, fc = ref.download()
    ix.run(reads, ref, fulgor_amd.THRESHOLD_UNION, 0.7)
    to, tc = ref.download()
    fo, to = fo.astype(np.int64), to.astype(np.int64)
    errors = []

    def worker(w):
        try:
            res = ix.new_result()
            for rep in range(3):
                for first in range((w * 1000) % 6000, 24000, 6000):
                    cnt = min(6000, 24000 - first)
                    algo = fulgor_amd.FULL_INTERSECTION if (w + rep) % 2 == 0 else fulgor_amd.THRESHOLD_UNION
                    ix.run(reads, res, algo, 0.7 if algo else 0.0, first, cnt)
                    go, gc = res.download()
                    wo, wc = (fo, fc) if algo == fulgor_amd.FULL_INTERSECTION else (to, tc)
                    if not (np.array_equal(go.astype(np.int64), wo[first:first + cnt + 1] - wo[first]) and
                            np.array_equal(gc, wc[wo[first]:wo[first + cnt]])):
                        errors.append((w, rep, first, int(algo)))
            res.close()
        except Exception as e:  # noqa: BLE001
            errors.append((w, repr(e)))

    ts = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert errors == []


@pytest.mark.parametrize("which", ["s10", "s4546"])
def test_device_compressed_formatter_parses_back_to_the_results(which, s10_gpu, seeded_reads, s4546):
    """psa_compressed_formatter on the device, built from the result bitmaps: the oracle's reader of the format must
    recover exactly the ids and colour lists of the pass (all three record kinds occur on s4546)"""
    from oracle.pyoracle import parse_compressed
    from fulgor_amd.driver import Formatter
    if which == "s10":
        ix, (b, o) = s10_gpu, seeded_reads
        b, o = b[:150 * 7000], o[:7001]
    else:
        ix, _, gen = s4546
        b, o = gen.generate(999, 7000, 150, 42)
    rd, res = ix.upload_reads(b, o), ix.new_result()
    for algo, tau, first in ((fulgor_amd.FULL_INTERSECTION, 0.0, 0), (fulgor_amd.THRESHOLD_UNION, 0.5, 4000000000)):
        ix.run(rd, res, algo, tau)
        offs, cols = res.download()
        data = Formatter("compressed", ix.num_colors()).header + bytes(res.format_view(2, first))
        ids, po, pc = parse_compressed(data)
        assert np.array_equal(ids, (np.arange(7000, dtype=np.uint64) + first).astype(np.uint32))
        assert np.array_equal(po, offs) and np.array_equal(pc, cols)
        if which == "s4546" and algo == fulgor_amd.FULL_INTERSECTION:
            sz, n = np.diff(offs.astype(np.int64)), ix.num_colors()
            assert (sz == 0).any() and ((sz > 0) & (sz < n // 4)).any() and ((sz >= n // 4) & (sz < 3 * n // 4)).any() and (sz >= 3 * n // 4 + 1).any()
        host = Formatter("compressed", ix.num_colors())
        assert len(data) < 1.2 * len(host.header + host.add(first, offs, cols) + host.finish()) + 64 * 7000 // 256


def test_kmer_tools_cli(s10_gpu, s10_fgidx, s10_oracle, tmp_path):
    """`fulgor kmer-conservation` / `kmer-matches` (tools/kmer_conservation.cpp, tools/kmer_matches.cpp): one line per
    record with its name; expected text assembled from the oracle's restatement of the two index members"""
    import subprocess, sys
    from conftest import ROOT
    reads = load_golden_reads()  # bytes
    sel = list(range(0, 40)) + list(range(1000, 1010))  # the edge cases (short, N, empty) are in 1000..1009
    fa = tmp_path / "q.fa"
    with open(fa, "w") as f:
        for i in sel:
            f.write(">q%d extra words\n%s\n" % (i, reads[i].decode()))
    k = s10_gpu.k()
    want_c, want_m = [], []
    prev_pos, prev_cnt = np.zeros(0, dtype=np.uint8), np.zeros(10, dtype=np.uint32)
    for i in sel:
        tr = s10_oracle.kmer_conservation(reads[i]) if len(reads[i]) >= k else []
        want_c.append("q%d\t%d%s\n" % (i, len(tr), "".join("\t(%d %d %d)" % t for t in tr)))
        if len(reads[i]) >= k:
            prev_pos, prev_cnt = s10_oracle.kmer_matches(reads[i])
        want_m.append("q%d\t%d%s%s\n" % (i, len(prev_pos), "".join("\t%d" % x for x in prev_pos), "".join("\t%d" % x for x in prev_cnt)))
    for tool, want in (("kmer-conservation", want_c), ("kmer-matches", want_m)):
        out = tmp_path / (tool + ".txt")
        rc = subprocess.run([sys.executable, "-m", "fulgor_amd", tool, "-i", s10_fgidx, "-q", str(fa), "-o", str(out), "-t", "2"],
                            cwd=ROOT).returncode
        assert rc == 0
        assert open(out).read() == "".join(want)
    assert subprocess.run([sys.executable, "-m", "fulgor_amd", "kmer-matches", "-i", s10_fgidx, "-q", str(tmp_path / "nope.fa"),
                           "-o", str(tmp_path / "x")], cwd=ROOT).returncode == 1


def test_s4546_config_size_10M_reads_properties(s4546):
    """BASELINE configs[2] / [3] at full size (10M reads, 4546 colours), through size-independent properties: the
    per-colour hit vector (a checksum over all 10M results) must not depend on how the reads are cut into passes;
    mapped counts agree; the union maps at least what the intersection maps; slices far apart inside the batch are bit-exact
    against the oracle; the compressed records of a pass parse back to its colour lists."""
    import torch
    from oracle.pyoracle import parse_compressed
    from fulgor_amd.driver import Formatter
    ix, orc, gen = s4546
    N = 10_000_000
    b, o = gen.generate(0, N, 150, 42)
    reads = ix.upload_reads(b, o)
    res = ix.new_result()
    nc = ix.num_colors()

    def hit_vector(algo, tau, chunk, expand=False):
        """expand: the u32 colour lists of every pass are built (k2b_expand, what bench.py times), the hit vector then comes from
        the expansion kernel's histogram, and the lists are checked by two independent device-side checksums"""
        hits = torch.zeros(nc + 2, dtype=torch.int64, device="cuda:0")
        total = mapped = 0
        for first in range(0, N, chunk):
            ix.run(reads, res, algo, tau, first, min(chunk, N - first))
            if expand:
                res.expand()
                from_lists, from_rows = res.checksum()
                # #entries of the CSR (its last offset, read on the device) = #set bits of the rows = the pass's total; sum and xor of
                # (colour + 1) * (position + 1) agree: every colour of every read is at its place
                assert from_lists == from_rows and from_lists[0] == res.sizes()[1] and from_lists[1] != 0
            res.accumulate_hits(hits.data_ptr())
            _, t, m = res.sizes()
            total += t
            mapped += m
        h = hits.cpu().numpy()
        assert h[nc] == N and h[nc + 1] == mapped and h[:nc].sum() == total
        return h

    fi_a = hit_vector(fulgor_amd.FULL_INTERSECTION, 0.0, 2_500_000)
    fi_b = hit_vector(fulgor_amd.FULL_INTERSECTION, 0.0, 1_300_000)   # ragged last pass
    assert np.array_equal(fi_a, fi_b)
    tu_a = hit_vector(fulgor_amd.THRESHOLD_UNION, 0.8, 2_500_000)
    tu_b = hit_vector(fulgor_amd.THRESHOLD_UNION, 0.8, 3_333_333)
    assert np.array_equal(tu_a, tu_b)
    # the same passes ending in the u32 colour lists (src/ps_full_intersection.cpp:376-400: the `colors` vector), as the bench times
    # them: the expansion kernel at config size, its histogram against the row-counting one, its lists against the rows
    ix.timing_enable(True)
    ix.timing_reset()
    assert np.array_equal(hit_vector(fulgor_amd.FULL_INTERSECTION, 0.0, 2_500_000, expand=True), fi_a)
    assert np.array_equal(hit_vector(fulgor_amd.THRESHOLD_UNION, 0.8, 2_500_000, expand=True), tu_a)
    assert ix.timing()["k2b_expand"][1] >= 8  # (eight launches of 2.5 M reads each)
    ix.timing_enable(False)
    # a read mapped by the full intersection has >= 1 positive k-mer, so the union at 0.8 maps it too; per colour the
    # union can only add reads
    assert tu_a[nc + 1] >= fi_a[nc + 1] and (tu_a[:nc] >= fi_a[:nc]).all()
    # slices far apart inside the big batch, bit-exact against the oracle
    for first in (0, 4_999_000, N - 20_000):
        cnt = 20_000
        lo, hi = int(o[first]), int(o[first + cnt])
        sb, so = b[lo:hi], o[first:first + cnt + 1] - o[first]
        ix.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, first, cnt)
        go, gc = res.download()
        oo, oc = orc.full_intersection(sb, so, threads=32)
        assert np.array_equal(go, oo) and np.array_equal(gc, oc)
        ids, po, pc = parse_compressed(Formatter("compressed", nc).header + bytes(res.format_view(2, first)))
        assert ids[0] == first and np.array_equal(po, go) and np.array_equal(pc, gc)
        ix.run(reads, res, fulgor_amd.THRESHOLD_UNION, 0.8, first, cnt)
        go, gc = res.download()
        oo, oc = orc.threshold_union(sb, so, 0.8, threads=32)
        assert np.array_equal(go, oo) and np.array_equal(gc, oc)


def test_s4546_config_metadiff_12M5_reads_per_gpu_properties(s4546):
    """BASELINE configs[4] at its PER-GPU size: meta-differential colour sets, 12.5 M reads (100 M reads over 8 GPUs), one GPU.
    Size-independent properties: the hit vector (what RCCL all-reduces) does not depend on how the reads are cut into passes
    and equals the hybrid index's on the same reads; slices far apart are bit-exact against the oracle's restated
    meta-differential cursors (merge_metadiff / meta_intersect<is_diff>)."""
    import torch
    from conftest import DATA
    from fulgor_amd import synth
    ix, orc_h, gen = s4546
    fg, _ = synth.ensure_s4546(DATA, S10_GENOMES)
    iy = fulgor_amd.Index(fg, device=0).convert(fulgor_amd.META_DIFF, 160, 16)
    iy.tune(dense_rows=False)  # the codec's own kernels (k_generic), not the dense rows
    N = 12_500_000
    b, o = gen.generate(3 * N, N, 150, 42)  # the slice rank 3 of the 8-GPU job owns
    nc = iy.num_colors()

    def hit_vector(index, chunk, expand=False):
        reads = index.upload_reads(b, o)
        res = index.new_result()
        hits = torch.zeros(nc + 2, dtype=torch.int64, device="cuda:0")
        for first in range(0, N, chunk):
            index.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, first, min(chunk, N - first))
            if expand:  # the u32 colour lists of the pass, checked against the rows by two independent device-side checksums
                res.expand()
                from_lists, from_rows = res.checksum()
                assert from_lists == from_rows and from_lists[0] == res.sizes()[1] and from_lists[1] != 0
            res.accumulate_hits(hits.data_ptr())
        res.close()
        reads.close()
        return hits.cpu().numpy()

    md_a = hit_vector(iy, 2_500_000)
    md_b = hit_vector(iy, 1_700_000)  # ragged last pass
    assert np.array_equal(md_a, md_b) and md_a[nc] == N
    iy.timing_enable(True)
    iy.timing_reset()
    assert np.array_equal(hit_vector(iy, 2_500_000, expand=True), md_a)  # (the expansion kernel's histogram against the row-counting one)
    assert iy.timing()["k2b_expand"][1] >= 5
    iy.timing_enable(False)
    assert np.array_equal(md_a, hit_vector(ix, 2_500_000))  # codecs do not change results
    from oracle.pyoracle import OracleIndex
    orc = OracleIndex.from_export(ix.export()).convert(fulgor_amd.META_DIFF, 160, 16)
    for first in (0, 6_000_000, N - 10_000):
        cnt = 10_000
        lo, hi = int(o[first]), int(o[first + cnt])
        sb, so = b[lo:hi], o[first:first + cnt + 1] - o[first]
        go, gc = iy.pseudoalign_full_intersection_batch(sb, so)
        oo, oc = orc.full_intersection(sb, so, threads=32)
        assert np.array_equal(go, oo) and np.array_equal(gc, oc)
        go, gc = iy.pseudoalign_threshold_union_batch(sb, so, 0.8)
        oo, oc = orc.threshold_union(sb, so, 0.8, threads=32)
        assert np.array_equal(go, oo) and np.array_equal(gc, oc)


@pytest.mark.parametrize("k", [21, 27])
def test_gpu_other_kmer_lengths(built, tmp_path, k):
    """k != 31 (the dictionary derives m = k - 12): a 3-genome index built at that k, lookups and both algorithms
    against the oracle. (k = 15, i.e. 3-base minimizers with very long overflow lists, also passes but takes minutes.)"""
    import subprocess
    from fulgor_amd.reads import ReadGenerator
    from oracle.pyoracle import OracleIndex
    base = str(tmp_path / ("s3_k%d" % k))
    subprocess.run([built.BIN_CCDBG, str(k), base] + S10_GENOMES[:3], check=True)
    ix = fulgor_amd.Index(base, device=0)
    assert ix.k() == k
    ix.selfcheck(unitig_stride=64)
    orc = OracleIndex.from_dump(base)
    b, o = ReadGenerator(S10_GENOMES[:3]).generate(0, 6000, 150, 3)
    i1, d1 = ix.fetch_color_set_ids_batch(b, o)
    i2, d2 = orc.fetch_color_set_ids(b, o, threads=16)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    go, gc = ix.pseudoalign_full_intersection_batch(b, o)
    oo, oc = orc.full_intersection(b, o, threads=16)
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)
    go, gc = ix.pseudoalign_threshold_union_batch(b, o, 0.6)
    oo, oc = orc.threshold_union(b, o, 0.6, threads=16)
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)


def test_s4546_reads_with_many_colour_sets(s4546):
    """chimeric long reads (40 fragments of 150 bp joined by N): a hundred and more distinct colour sets per read, i.e.
    several descriptor groups per read in both kernels (k3a takes lists in groups of 16, k2a of 64), 16-bit score counters"""
    ix, orc, gen = s4546
    b, o = gen.generate(424242, 40 * 60, 150, 42)
    frags = [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1)]
    reads = [b"N".join(frags[40 * j:40 * j + 40]) for j in range(60)] + [b"N".join(frags[:7]), frags[3]]
    bb, oo_ = pack_reads(reads)
    i1, d1 = ix.fetch_color_set_ids_batch(bb, oo_)
    assert np.diff(i1.astype(np.int64)).max() > 64
    for tau in (0.05, 0.4):
        go, gc = ix.pseudoalign_threshold_union_batch(bb, oo_, tau)
        wo, wc = orc.threshold_union(bb, oo_, tau, threads=16)
        assert np.array_equal(go, wo) and np.array_equal(gc, wc)
    go, gc = ix.pseudoalign_full_intersection_batch(bb, oo_)
    wo, wc = orc.full_intersection(bb, oo_, threads=16)
    assert np.array_equal(go, wo) and np.array_equal(gc, wc)


def test_gpu_hit_counts_with_capped_blocks(s10_fgidx):
    """the expand kernel keeps its per-colour hit histogram in 16-bit LDS counters, so a block must never take
    more than 65535 reads although tickets are handed out dynamically. Blocks stop pulling at a cap; with the cap lowered to
    64 reads (test knob, read once per process, hence the subprocess) most blocks hit it on a 60000-read pass, and the hit
    vector must still equal the column sums of the downloaded results."""
    import subprocess
    code = r'''
import glob, os, sys
import numpy as np, torch
sys.path.insert(0, %r)
import fulgor_amd
from fulgor_amd.reads import ReadGenerator
g = sorted(glob.glob(os.path.join(%r, "tests", "data", "salmonella_10", "*.fasta.gz")))
b, o = ReadGenerator(g).generate(7, 60000, 150, 11)
ix = fulgor_amd.Index(%r, device=0)
reads = ix.upload_reads(b, o)
res = ix.new_result()
n = ix.num_colors()
for algo, tau in ((fulgor_amd.FULL_INTERSECTION, 0.0), (fulgor_amd.THRESHOLD_UNION, 0.5)):
    ix.run(reads, res, algo, tau)
    res.expand()  # (the histogram under test is the one the expand kernel keeps; without the lists asked for, k_hits counts from the rows)
    hits = torch.zeros(n + 2, dtype=torch.int64, device="cuda:0")
    res.accumulate_hits(hits.data_ptr())
    go, gc = res.download()
    got = hits.cpu().numpy()
    assert np.array_equal(got[:n], np.bincount(gc, minlength=n)), "hit vector differs"
    assert got[n] == 60000 and got[n + 1] == int((np.diff(go.astype(np.int64)) > 0).sum())
print("ok")
''' % (ROOT, ROOT, s10_fgidx)
    env = dict(os.environ, FULGOR_EXPAND_BLOCK_CAP="64")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_colour_lists_in_the_fastest_of_several_allocations(s4546):
    """FULGOR_EXPAND_LOTTERY=2 (off by default; read once per process, hence the subprocess): a result's colour lists of 2 GB or more
    are allocated up to three times when they are first sized, the expansion kernel is timed on each (as k_probe_allocation) and the
    fastest allocation is kept. The lists it ends up with carry the same checksum as the result rows, and the hit vector equals the
    row-counting one."""
    import subprocess
    from conftest import DATA
    fg = os.path.join(DATA, "s4546syn.v9.fgidx")
    code = r'''
import glob, os, sys
import numpy as np, torch
sys.path.insert(0, %r)
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import ReadGenerator
g = sorted(glob.glob(os.path.join(%r, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(%r, g)
gen = ReadGenerator(g, raw_sequences=extra)
n = 1200000
b, o = gen.generate(5000000, n, 150, 21)
ix = fulgor_amd.Index(fg, device=0)
ix.timing_enable(True)
reads = ix.upload_reads(b, o)
res = ix.new_result()
nc = ix.num_colors()
for algo, tau in ((fulgor_amd.FULL_INTERSECTION, 0.0), (fulgor_amd.THRESHOLD_UNION, 0.8)):
    ix.run(reads, res, algo, tau)
    lazy = torch.zeros(nc + 2, dtype=torch.int64, device="cuda:0")
    res.accumulate_hits(lazy.data_ptr())  # (from the rows: nobody has asked for the lists yet)
    res.expand()
    from_lists, from_rows = res.checksum()
    assert from_lists == from_rows and from_lists[0] == res.sizes()[1] > 0, (from_lists, from_rows)
    hist = torch.zeros(nc + 2, dtype=torch.int64, device="cuda:0")
    res.accumulate_hits(hist.data_ptr())
    assert torch.equal(hist, lazy)
assert ix.timing()["k2b_expand"][1] == 2, ix.timing()["k2b_expand"]  # (the timing runs are not this kernel's launches)
print("ok")
''' % (ROOT, ROOT, DATA)
    env = dict(os.environ, FULGOR_EXPAND_LOTTERY="2", FULGOR_TRACE_ALLOC="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stderr.count("[lottery] kept") >= 1 and "[lottery] candidate 2" in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("cu_split", ["", "96"])
def test_gpu_lookup_and_colour_stage_as_two_calls(s10_fgidx, cu_split):
    """fgpu_run_lookup + fgpu_run_colours (a worker loop that keeps the lookup of the next batch in flight beside the colour stage
    of the current one: two results, alternating) give what fgpu_run gives, batch by batch — on one stream, and with
    FULGOR_CU_SPLIT (lookup kernels and colour kernels on disjoint sets of CUs, ordered by an event; read when a result is
    created, hence the subprocess)."""
    import subprocess
    code = r'''
import glob, os, sys
import numpy as np, torch
sys.path.insert(0, %r)
import fulgor_amd
from fulgor_amd.reads import ReadGenerator
g = sorted(glob.glob(os.path.join(%r, "tests", "data", "salmonella_10", "*.fasta.gz")))
b, o = ReadGenerator(g).generate(3, 50000, 150, 5)
ix = fulgor_amd.Index(%r, device=0)
reads = ix.upload_reads(b, o)
ref, two = ix.new_result(), [ix.new_result(), ix.new_result()]
n = ix.num_colors()
chunks = [(0, 20000), (20000, 17000), (37000, 13000)]
for algo, tau in ((fulgor_amd.FULL_INTERSECTION, 0.0), (fulgor_amd.THRESHOLD_UNION, 0.7)):
    want = []
    for first, cnt in chunks:
        ix.run(reads, ref, algo, tau, first, cnt)
        h = torch.zeros(n + 2, dtype=torch.int64, device="cuda:0")
        ref.accumulate_hits(h.data_ptr())
        want.append(ref.download() + (h.cpu().numpy(),))
    ix.run_lookup(reads, two[0], chunks[0][0], chunks[0][1])
    for t in range(len(chunks)):
        if t + 1 < len(chunks):
            ix.run_lookup(reads, two[(t + 1) & 1], chunks[t + 1][0], chunks[t + 1][1])  # in flight beside the colour stage below
        ix.run_colours(two[t & 1], algo, tau)
        h = torch.zeros(n + 2, dtype=torch.int64, device="cuda:0")
        two[t & 1].accumulate_hits(h.data_ptr())
        got = two[t & 1].download() + (h.cpu().numpy(),)
        assert all(np.array_equal(a, c) for a, c in zip(want[t], got)), "pass %%d differs" %% t
print("ok")
''' % (ROOT, ROOT, s10_fgidx)
    env = dict(os.environ)
    if cu_split:
        env["FULGOR_CU_SPLIT"] = cu_split
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


# ---- round 2: parity evidence without shared inputs ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def s4546small(built):
    """4546 colours, list shapes of the big synthetic index, small enough to travel as text: the index is written with
    `fulgor dump`'s format (fgpu_dump) and the oracle builds itself from those files, ENCODING the colour sets with its own
    encoder: the two sides share no encoded stream and no unitig table."""
    from conftest import DATA
    from fulgor_amd import synth
    from fulgor_amd.reads import ReadGenerator
    from oracle.pyoracle import OracleIndex
    fg, extra = synth.ensure_s4546_small(DATA, S10_GENOMES)
    ix = fulgor_amd.Index(fg, device=0)
    base = os.path.join(DATA, "s4546small_dump")
    if not os.path.exists(base + ".unitigs.fa"):
        ix.dump(base)
    orc = OracleIndex.from_dump(base)
    gen = ReadGenerator(S10_GENOMES[:1], raw_sequences=extra)
    return ix, orc, gen, fg, base


def test_s4546small_dump_feeds_the_oracle(s4546small):
    ix, orc, gen, _, _ = s4546small
    ex = ix.export()
    words, offs = orc.encoded_colors()  # encoded by the oracle from the text lists
    assert np.array_equal(offs, ex["color_offsets"])
    assert np.array_equal(words, ex["color_words"][:len(words)])
    b, o = gen.generate(0, 20000, 150, 42)
    i1, d1 = ix.fetch_color_set_ids_batch(b, o)
    i2, d2 = orc.fetch_color_set_ids(b, o, threads=32)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    assert int(np.diff(i1.astype(np.int64)).max()) >= 3  # reads do cross unitigs of different colour sets
    go, gc = ix.pseudoalign_full_intersection_batch(b, o)
    oo, oc = orc.full_intersection(b, o, threads=32)
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)
    for tau in (0.8, 0.3):
        go, gc = ix.pseudoalign_threshold_union_batch(b, o, tau)
        oo, oc = orc.threshold_union(b, o, tau, threads=32)
        assert np.array_equal(go, oo) and np.array_equal(gc, oc), tau


@pytest.mark.parametrize("index_type,psize,csize", [(fulgor_amd.DIFF, 4546, 16), (fulgor_amd.META, 160, 1), (fulgor_amd.META_DIFF, 160, 16)])
def test_s4546small_codecs_equal_oracle_convert(s4546small, index_type, psize, csize):
    """meta / differential / meta-differential at 4546 colours DIRECTLY against the oracle's restated cursors and
    meta_intersect / diff_intersect / merge_* (round 1 compared them with the hybrid HIP path only)"""
    from oracle.pyoracle import OracleIndex
    _, _, gen, fg, base = s4546small
    iy = fulgor_amd.Index(fg, device=0).convert(index_type, psize, csize)
    iy.tune(dense_rows=False)  # the codec's own kernels (k_generic), not the dense rows
    orc = OracleIndex.from_dump(base).convert(index_type, psize, csize)
    b, o = gen.generate(50000, 12000, 150, 42)
    go, gc = iy.pseudoalign_full_intersection_batch(b, o)
    oo, oc = orc.full_intersection(b, o, threads=32)
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)
    go, gc = iy.pseudoalign_threshold_union_batch(b, o, 0.8)
    oo, oc = orc.threshold_union(b, o, 0.8, threads=32)
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)


@pytest.mark.parametrize("index_type,psize,csize", [(fulgor_amd.HYBRID, 0, 0), (fulgor_amd.DIFF, 16, 4), (fulgor_amd.META, 48, 8), (fulgor_amd.META_DIFF, 48, 8)])
def test_gpu_matches_golden_at_256_colours(c256_dump, index_type, psize, csize, colour_stage):
    """golden vectors ABOVE 64 colours: computed by the independent k-mer oracle straight from the 256 seeded genomes
    (tests/golden/make_golden_c256.py); nothing of the restatement is involved"""
    ix = fulgor_amd.Index(c256_dump, device=0)
    if index_type != fulgor_amd.HYBRID:
        ix.convert(index_type, psize, csize)
    ix.tune(dense_rows=colour_stage)  # False: the codec's own kernels (k2a / k3a on the hybrid lists, k_generic on the others)
    b, o = pack_reads(load_golden_reads("c256_reads.fa"))
    offs, cols = ix.pseudoalign_full_intersection_batch(b, o)
    assert csr_to_lists(offs, cols) == load_golden_tsv("c256_full_intersection.tsv")
    for tau in (0.8, 0.3):
        offs, cols = ix.pseudoalign_threshold_union_batch(b, o, tau)
        assert csr_to_lists(offs, cols) == load_golden_tsv("c256_threshold_union_%s.tsv" % tau)


def test_cli_two_ranks_share_the_gpu(s10_fgidx, s10_oracle, tmp_path):
    """`pseudoalign --gpus 2` (own launcher, one process per rank; FULGOR_SHARE_GPU=1 puts both ranks on the one GPU of this
    box and the counters on gloo): the joined output equals the single-process output byte for byte (ascii), parses back to
    the same records (compressed), and the pipelined single-process output equals the oracle's."""
    import subprocess
    from fulgor_amd.reads import ReadGenerator
    from oracle.pyoracle import parse_compressed
    n = 150_000
    b, o = ReadGenerator(S10_GENOMES).generate(9, n, 150, 5)
    q = tmp_path / "reads.fq"
    with open(q, "wb") as f:
        for i in range(n):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, bytes(b[int(o[i]):int(o[i + 1])]), b"I" * 150))
    base = [sys.executable, "-m", "fulgor_amd", "pseudoalign", "-i", s10_fgidx, "-q", str(q), "--verbose"]
    outs = {}
    for fmt in ("ascii", "compressed"):
        for gpus in (1, 2):
            out = tmp_path / ("out_%s_%d" % (fmt, gpus))
            env = dict(os.environ, FULGOR_SHARE_GPU="1")
            r = subprocess.run(base + ["-o", str(out), "--format", fmt, "--gpus", str(gpus)], cwd=ROOT, env=env,
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout + r.stderr
            assert "processed %d reads" % n in r.stdout
            outs[fmt, gpus] = open(out, "rb").read()
    assert outs["ascii", 1] == outs["ascii", 2]
    oo, oc = s10_oracle.full_intersection(b, o, threads=32)
    assert outs["ascii", 1] == s10_oracle.format_ascii(oo, oc)
    for gpus in (1, 2):
        ids, po, pc = parse_compressed(outs["compressed", gpus])
        assert np.array_equal(ids, np.arange(n)) and np.array_equal(po, oo) and np.array_equal(pc, oc)


@pytest.mark.parametrize("suffix,index_type,psize,csize", [("fur", 0, 0, 0), ("mdfur", 3, 4, 2)])
def test_gpu_queries_on_an_index_loaded_from_the_fur_layout(s10_fgidx, s10_oracle, seeded_reads, tmp_path, suffix, index_type, psize, csize):
    """an index written in the reference's section layout (fur_format.hpp; own k2u block) and opened again answers like the
    index it was written from"""
    ix = fulgor_amd.Index(s10_fgidx, device=-1)
    if index_type:
        ix.convert(index_type, psize, csize)
    p = str(tmp_path / ("x." + suffix))
    ix.save(p)
    iy = fulgor_amd.Index(p, device=0)
    if index_type:
        iy.tune(dense_rows=False)  # the codec's own kernels (k_generic), not the dense rows
    assert iy.index_type == index_type
    b, o = seeded_reads
    b, o = b[:int(o[8000])], o[:8001]
    go, gc = iy.pseudoalign_full_intersection_batch(b, o)
    oo, oc = s10_oracle.full_intersection(b, o, threads=16)
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)
    go, gc = iy.pseudoalign_threshold_union_batch(b, o, 0.8)
    oo, oc = s10_oracle.threshold_union(b, o, 0.8, threads=16)
    assert np.array_equal(go, oo) and np.array_equal(gc, oc)


def test_rccl_all_reduce_of_the_hit_vector_single_rank(s10_fgidx, seeded_reads, tmp_path):
    """the reduction `bench.py` and `pseudoalign --gpus N` run at the end of a step, on the backend they use on a multi-GPU
    node ("nccl" = RCCL on ROCm), with the one rank this box offers: process-group set-up on the device, all-reduce of the
    device-resident hit vector that fgpu_result_accumulate_hits filled, result unchanged. (What a single GPU can show of
    the RCCL path; the ranks' control flow is covered on gloo.)"""
    import subprocess
    code = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import fulgor_amd
from fulgor_amd import driver
from fulgor_amd.reads import ReadGenerator
import glob
g = sorted(glob.glob(os.path.join(%r, "tests", "data", "salmonella_10", "*.fasta.gz")))
b, o = ReadGenerator(g).generate(0, 20000, 150, 7)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ix = fulgor_amd.Index(%r, device=0)
reads = ix.upload_reads(b, o)
res = ix.new_result()
ix.run(reads, res, fulgor_amd.FULL_INTERSECTION)
n = ix.num_colors()
hits = torch.zeros(n + 2, dtype=torch.int64, device="cuda:0")
res.accumulate_hits(hits.data_ptr())
before = hits.cpu().numpy().copy()
driver.all_reduce_hits(hits)
dist.barrier()
torch.cuda.synchronize()
go, gc = res.download()
assert np.array_equal(hits.cpu().numpy(), before) and np.array_equal(before[:n], np.bincount(gc, minlength=n))
dist.destroy_process_group()
print("ok")
''' % (ROOT, ROOT, s10_fgidx)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout.split(), r.stdout + r.stderr  # (the runtime may print after the script's last line)


@pytest.mark.gpu
def test_pooled_buffers_and_reused_stream_results(s10_gpu, s10_oracle, tmp_path):
    """The command-line loop recycles what it used to allocate per batch: device read buffers (pool in the index), pinned
    reader buffers (process-wide pool), the three results of the stream (kept with the index). Batches of growing and shrinking
    size through the same index, and two streams over two files one after the other, must give the oracle's records."""
    from fulgor_amd import driver
    from fulgor_amd.reads import FastxReader, ReadGenerator
    from conftest import S10_GENOMES
    gen = ReadGenerator(S10_GENOMES)
    for n, seed in ((3000, 5), (200, 6), (9000, 7), (1, 8), (4000, 9)):  # upload / free / upload: larger, smaller, larger again
        bases, offs = gen.generate(0, n, 150, seed)
        reads = s10_gpu.upload_reads(bases, offs)
        res = s10_gpu.new_result()
        s10_gpu.run(reads, res, 0, 0.0)
        o, c = res.download()
        reads.close()
        res.close()
        wo, wc = s10_oracle.full_intersection(bases, offs)
        assert np.array_equal(np.asarray(o, dtype=np.int64), np.asarray(wo, dtype=np.int64)) and np.array_equal(c, wc)
    outs = []
    for n, seed in ((7000, 11), (2500, 12)):
        bases, offs = gen.generate(0, n, 150, seed)
        fq = tmp_path / ("q%d.fq" % seed)
        with open(fq, "wb") as f:
            for i in range(n):
                s = bytes(bases[int(offs[i]):int(offs[i + 1])])
                f.write(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)))
        out = tmp_path / ("o%d.tsv" % seed)
        rd = FastxReader(str(fq), batch=1024, copy=False, threads=3)
        with open(out, "wb") as sink:
            got, mapped = driver.pseudoalign_stream(s10_gpu, rd, sink=sink, fmt="ascii")
        rd.close()
        assert got == n
        wo, wc = s10_oracle.full_intersection(bases, offs)
        wo = np.asarray(wo, dtype=np.int64)
        want = "".join("%d\t%d%s\n" % (i, wo[i + 1] - wo[i], "".join("\t%d" % x for x in wc[wo[i]:wo[i + 1]])) for i in range(n))
        assert out.read_bytes().decode() == want
        outs.append(mapped)
    assert len(getattr(s10_gpu, "_stream_results", [])) == 3  # the second stream found the results of the first


def test_one_cold_command_is_prepared_timed_like_the_reference_and_leaves_fast(s10_fgidx, s10_oracle, tmp_path):
    """round-5 review, item 2. (1) The one-run preparation (fgpu_prepare_host + fgpu_stream_prepare: host buffers pinned, worker
    results created ahead of the first batch) changes no byte: a prepared stream, an unprepared one, a prepared one on a tiny file, one
    prepared for another format and for shorter reads than come (buffers grow) all write the oracle's records; no host buffer is
    pinned during a prepared run. (2) The command line says when the index is loaded and starts its `elapsed` clock behind the load,
    where the reference starts its own (tools/pseudoalign.cpp:59-60): elapsed is far less than the wall of the command; with and
    without the preparation, leaving through _exit or through the interpreter (FULGOR_ORDERLY_EXIT), the output is the same file."""
    import subprocess
    import time
    from fulgor_amd.index import prepare_host
    from fulgor_amd.reads import FastxReader, ReadGenerator
    gen = ReadGenerator(S10_GENOMES)
    n = 60000
    b, o = gen.generate(77, n, 150, 3)
    q = tmp_path / "reads.fq"
    with open(q, "wb") as f:
        for i in range(n):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, bytes(b[int(o[i]):int(o[i + 1])]), b"I" * 150))
    oo, oc = s10_oracle.full_intersection(b, o, threads=32)
    want = s10_oracle.format_ascii(oo, oc)
    ix = fulgor_amd.Index(s10_fgidx, device=0)
    prepare_host(0, reader_threads=3, workers=3, batch=8192, text_bytes_per_read=316, fastq=True, out_bytes_per_read=64, total_text_bytes=os.path.getsize(q))
    ix.stream_prepare(0, 8192, 3, 100, 64)  # (ascii, reads of up to 100 bases: the 150-base reads that come make the buffers grow)
    for fmt in (0, 2):
        out = tmp_path / ("stream_%d" % fmt)
        rd = FastxReader(str(q), batch=8192, copy=False, threads=3)
        with open(out, "wb") as sink:
            got, mapped = ix.pseudoalign_stream(rd, sink.fileno(), 0, 0.0, fmt, 0, True, 8192, 3)
        rd.close()
        assert got == n
        if fmt == 0:
            assert out.read_bytes() == want
        else:
            from oracle.pyoracle import parse_compressed
            ids, po, pc = parse_compressed(out.read_bytes())
            assert np.array_equal(ids, np.arange(n)) and np.array_equal(po, oo) and np.array_equal(pc, oc)
    tiny = tmp_path / "tiny.fq"
    tiny.write_bytes(b"".join(open(q, "rb").read(316 * 8).splitlines(True)[:28]))  # seven records
    prepare_host(0, text_bytes_per_read=316, fastq=True, out_bytes_per_read=256, total_text_bytes=os.path.getsize(tiny))
    rd = FastxReader(str(tiny), batch=1 << 18, copy=False, threads=2)
    with open(tmp_path / "tiny.out", "wb") as sink:
        got, _ = ix.pseudoalign_stream(rd, sink.fileno(), 0, 0.0, 0, 0, True, 0, 0)
    rd.close()
    assert got == 7 and (tmp_path / "tiny.out").read_bytes() == b"".join(want.splitlines(True)[:7])
    ix.close()
    # (2) the command line
    outs = {}
    for tag, env_extra in (("prepared", {}), ("unprepared", {"FULGOR_NO_PREPARE": "1"}), ("orderly", {"FULGOR_ORDERLY_EXIT": "1"})):
        out = tmp_path / ("cli_" + tag)
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "fulgor_amd", "pseudoalign", "-i", s10_fgidx, "-q", str(q), "-o", str(out), "--verbose"],
                           cwd=ROOT, env=dict(os.environ, FULGOR_CLI_TIMELINE="%.6f" % t0, **env_extra), capture_output=True, text=True, timeout=600)
        wall = time.time() - t0
        assert r.returncode == 0, r.stdout + r.stderr
        lines = r.stdout.splitlines()
        i_start, i_done = [i for i, l in enumerate(lines) if "*** START: loading the index" in l], [i for i, l in enumerate(lines) if "*** DONE: loading the index" in l]
        assert i_start and i_done and i_start[0] < i_done[0] < [i for i, l in enumerate(lines) if l.startswith("processed %d reads" % n)][0]
        open_ms = int(lines[i_done[0]].split("(")[1].split()[0])
        elapsed_ms = int([l for l in lines if l.startswith("elapsed = ")][0].split()[2])
        assert elapsed_ms + open_ms <= wall * 1000 + 5 and elapsed_ms < 0.6 * wall * 1000, (elapsed_ms, open_ms, wall)  # the clock does not cover the load
        outs[tag] = out.read_bytes()
    assert outs["prepared"] == want and outs["unprepared"] == want and outs["orderly"] == want
    # compressed records (what the preparation sizes the output buffers for): nothing is pinned during the prepared run
    r = subprocess.run([sys.executable, "-m", "fulgor_amd", "pseudoalign", "-i", s10_fgidx, "-q", str(q), "-o", str(tmp_path / "cli_c"), "--format", "compressed"],
                       cwd=ROOT, env=dict(os.environ, FULGOR_CLI_TIMELINE="%.6f" % time.time()), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "host buffers pinned anew during the run: 0 " in r.stderr, r.stderr[-1500:]


def test_s4546_execution_knobs_do_not_change_results(s4546):
    """fgpu_tune: dense rows (k2r_intersect) or packed blocks (k2a_intersect), the locality order of a pass (k_order_*: reads
    sorted by their rarest colour set, results at the read's own row) and the small-result bypass (results of at most 16 colours travel from k2a to k2b as colours, no bitmap row) are
    execution choices — offsets, colours, hit counts and all three output formats must be the same bytes in every combination,
    and equal to the oracle's"""
    import torch
    from oracle.pyoracle import parse_compressed
    from fulgor_amd.driver import Formatter
    ix, orc, gen = s4546
    n = 40000
    b, o = gen.generate(777000, n, 150, 42)
    # a few reads on top whose results are forced small / empty / dense through their position in the batch do not exist:
    # the generator's mix (5 % random reads, accessory and core loci) holds all of them, checked below
    rd, res = ix.upload_reads(b, o), ix.new_result()
    ncol = ix.num_colors()
    outs = []
    try:
        for order_min, small, rows in ((-1, False, False), (1, False, False), (-1, True, False), (1, True, False),
                                       (-1, False, True), (1, True, True)):
            ix.tune(order_min_reads=order_min, small_results=small, dense_rows=rows)
            ix.run(rd, res, fulgor_amd.FULL_INTERSECTION)
            offs, cols = res.download()
            hits = torch.zeros(ncol + 2, dtype=torch.int64, device="cuda:0")
            res.accumulate_hits(hits.data_ptr())
            outs.append((offs, cols, hits.cpu().numpy(), bytes(res.format_view(0, 5)), bytes(res.format_view(1, 5)),
                         bytes(res.format_view(2, 5))))
            ix.run(rd, res, fulgor_amd.THRESHOLD_UNION, 0.8)  # (the union kernels under the same knobs)
            outs[-1] += res.download()
    finally:
        ix.tune(order_min_reads=-1, small_results=True, dense_rows=True)
    sz = np.diff(outs[0][0].astype(np.int64))
    assert (sz == 0).any() and ((sz > 0) & (sz <= 16)).sum() > n // 10 and (sz > 16).sum() > n // 10
    for other in outs[1:]:
        for a, c in zip(outs[0], other):
            assert np.array_equal(a, c) if isinstance(a, np.ndarray) else a == c
    offs, cols, hits = outs[0][0], outs[0][1], outs[0][2]
    oo, oc = orc.full_intersection(b, o, threads=32)
    assert np.array_equal(offs, oo) and np.array_equal(cols, oc)
    assert np.array_equal(hits[:ncol], np.bincount(cols, minlength=ncol)) and hits[ncol] == n and hits[ncol + 1] == (sz > 0).sum()
    ids, po, pc = parse_compressed(Formatter("compressed", ncol).header + outs[5][5])
    assert np.array_equal(ids, np.arange(5, 5 + n, dtype=np.uint32)) and np.array_equal(po, offs) and np.array_equal(pc, cols)
    assert outs[5][3] == Formatter("ascii", ncol).add(5, offs, cols)
    uo, uc = orc.threshold_union(b, o, 0.8, threads=32)
    assert np.array_equal(outs[5][6], uo) and np.array_equal(outs[5][7], uc)


def test_bench_starts_its_own_ranks_and_checks_the_reduction(tmp_path):
    """`python bench.py --gpus 2` outside any launcher: it starts its two ranks itself (both on the one GPU of the test box, gloo:
    FULGOR_BENCH_SHARE_GPU=1), every rank processes its own reads, the hit vector is all-reduced and the bench asserts that the
    reads of all ranks are in it; the line carries the rank count, the backend and a roofline entry per kernel"""
    import json
    import subprocess
    env = dict(os.environ, FULGOR_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "s10", "--reads", "200000", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo" and line["scaling"] == "weak"
    assert line["value"] > 0 and line["steps"] == 2 and line["config"]["reads_per_gpu"] == 200000
    ks = line["roofline"]["kernels"]
    assert set(ks) == {"k1_lookup", "k2_intersect", "k2b_expand"}
    assert all(0 < v["frac"] < 1 and v["ms"] > 0 and v["GB"] > 0 for v in ks.values())
    assert line["roofline"]["kernel"] in ks and abs(line["roofline"]["frac"] - ks[line["roofline"]["kernel"]]["frac"]) < 1e-4
    assert len(r.stdout.strip().splitlines()[-1]) < 8000  # (the driver keeps 9 KB of stdout: the line must fit)
    assert [x[0] for x in line["ranks"]] == [0, 1] and "0x" in line["ranks"][0][3]  # every rank's device, free memory and copy engines
    detail = json.load(open(os.path.join(ROOT, line["detail"])))  # everything that was measured, beside the compact line
    assert detail["value"] == line["value"] and "avg_launch_ms" in detail["roofline"]["kernels"]["k1_lookup"]


def test_eight_ranks_share_the_gpu_command_line_and_bench(s10_fgidx, s10_oracle, tmp_path):
    """round-5 review, item 3: what an 8-GPU node runs first, on the one GPU of this box — `pseudoalign --gpus 8` and `bench.py --gpus 8`
    with all eight ranks on cuda:0 (FULGOR_SHARE_GPU / FULGOR_BENCH_SHARE_GPU, counters over gloo): eight processes open the index,
    time the copy engines (under the host-wide lock) and stream their part at once; the joined ascii output is the single-rank
    output byte for byte and the oracle's; every rank reports its device and engines; the reduction counts the reads of all eight."""
    import json
    import subprocess
    from fulgor_amd.reads import ReadGenerator
    n = 120_000
    b, o = ReadGenerator(S10_GENOMES).generate(3, n, 150, 8)
    q = tmp_path / "reads.fq"
    with open(q, "wb") as f:
        for i in range(n):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, bytes(b[int(o[i]):int(o[i + 1])]), b"I" * 150))
    outs = {}
    for gpus in (1, 8):
        out = tmp_path / ("out_%d" % gpus)
        env = dict(os.environ, FULGOR_SHARE_GPU="1", FULGOR_TRACE_OPENS="1")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, "-m", "fulgor_amd", "pseudoalign", "-i", s10_fgidx, "-q", str(q), "-o", str(out), "--format", "ascii",
                            "--gpus", str(gpus), "--verbose"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "processed %d reads" % n in r.stdout
        outs[gpus] = open(out, "rb").read()
        if gpus == 8:  # one open of the query file and one of the index per rank, and every rank says where it runs
            assert r.stderr.count("[rank] query part opened") == 8 and r.stderr.count("[rank] index opened") == 8, r.stderr[-3000:]
            assert all(("[rank %d/8]" % k) in r.stderr for k in range(8)), r.stderr[-3000:]
            assert not [p for p in os.listdir(tmp_path) if ".part" in p]
    assert outs[1] == outs[8]
    oo, oc = s10_oracle.full_intersection(b, o, threads=32)
    assert outs[1] == s10_oracle.format_ascii(oo, oc)
    env = dict(os.environ, FULGOR_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "s10", "--reads", "200000", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["reads_counted_all_ranks"] == 8 * 200000
    assert [x[0] for x in line["ranks"]] == list(range(8))
    assert r.stderr.count("[bench] rank ") >= 16  # (two lines per rank: device and memory before the workload, the device report after the open)


@pytest.mark.parametrize("index_type,psize,csize", [(fulgor_amd.DIFF, 4546, 16), (fulgor_amd.META, 160, 1), (fulgor_amd.META_DIFF, 160, 16)])
def test_s4546_dense_rows_serve_every_codec(s4546, index_type, psize, csize):
    """the dense rows are built from the colour sets themselves, so an index re-encoded as differential / meta / meta-differential
    answers from them too (k2r_intersect / k3r_union) unless told otherwise: the same results as the codec's own kernels and as
    the hybrid index"""
    from conftest import DATA
    from fulgor_amd import synth
    ix, _, gen = s4546
    fg, _ = synth.ensure_s4546(DATA, S10_GENOMES)
    b, o = gen.generate(4242, 12000, 150, 42)
    iy = fulgor_amd.Index(fg, device=0).convert(index_type, psize, csize)
    want = ix.pseudoalign_full_intersection_batch(b, o), ix.pseudoalign_threshold_union_batch(b, o, 0.7)
    for rows in (True, False):
        iy.tune(dense_rows=rows)
        got = iy.pseudoalign_full_intersection_batch(b, o), iy.pseudoalign_threshold_union_batch(b, o, 0.7)
        for (wo, wc), (go, gc) in zip(want, got):
            assert np.array_equal(wo, go) and np.array_equal(wc, gc), rows
    iy.close()


@pytest.mark.parametrize("index_type", ["hybrid", "meta-diff"])
def test_bench_two_ranks_on_the_synthetic_4546_colour_index(index_type):
    """(meta-diff: the codec of configs[4], the 8-GPU configuration, on two ranks at reduced size)
    first contact of the multi-GPU bench (the driver's 8-GPU run) as far as one GPU can show it: `bench.py --gpus 2` on the
    BASELINE workload itself — rank 0 builds the synthetic 4546-colour index (tens of seconds on a cold box) while rank 1 waits
    for the marker file, both open it, every rank announces its device on stderr, the hit vector is all-reduced and checked"""
    import json
    import subprocess
    env = dict(os.environ, FULGOR_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FULGOR_S4546_DUMP"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "s4546syn", "--reads", "200000", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--index-type", index_type], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo" and line["scaling"] == "weak"
    assert line["reads_counted_all_ranks"] == 2 * 200000 and 0 < line["mapped_all_ranks"] <= 2 * 200000
    assert ("meta-diff" in line["config"]["workload"]) == (index_type == "meta-diff")
    assert line["data"] == "synthetic" and "SYNTHETIC" in line["config"]["workload"] and line["config"]["reads_per_gpu"] == 200000
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1
    for rank in (0, 1):
        assert "[bench] rank %d/2 on cuda:0" % rank in r.stderr


def test_bench_on_a_real_dump_says_so(s4546small):
    """FULGOR_S4546_DUMP: the bench line of an index ingested from `fulgor dump` files carries "data": "real-dump" and the dump's
    name (here the dump is the small 4546-colour test index written by fgpu_dump)"""
    import json
    import subprocess
    _, _, _, _, base = s4546small
    env = dict(os.environ, FULGOR_S4546_DUMP=base)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "20000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--no-secondary"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["data"] == "real-dump" and os.path.basename(base) in line["config"]["workload"] and "REAL DUMP" in line["config"]["workload"]
    assert line["config"]["reads_per_gpu"] == 20000 and line["value"] > 0 and line["config"]["mapped_fraction"] > 0.8


def test_bench_pipelined_passes_count_every_read():
    """`bench.py --pipeline 1`: the passes of all steps as one sequence, the lookup of pass t + 1 queued before the colour stage of pass t
    is waited for (fgpu_run_lookup / fgpu_run_colours on two results). The line must account for the same reads, mapped reads and
    colours as the plain loop over the same passes."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    lines = []
    for pipe in ("0", "1"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "s10", "--reads", "150000", "--chunk", "40000", "--steps", "2",
                            "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--pipeline", pipe], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines.append(json.loads(r.stdout.strip().splitlines()[-1]))
        lines[-1]["_detail"] = json.load(open(os.path.join(ROOT, lines[-1]["detail"])))  # (the launch counts are in the detail file beside the compact line)
    a, b = lines
    assert a["config"]["reads_per_gpu"] == b["config"]["reads_per_gpu"] == 150000 and b["value"] > 0 and b["steps"] == 2
    assert a["config"]["mapped_fraction"] == b["config"]["mapped_fraction"] and a["config"]["avg_colours_per_read"] == b["config"]["avg_colours_per_read"]
    assert b["_detail"]["kernels"]["k1_lookup"]["launches"] == a["_detail"]["kernels"]["k1_lookup"]["launches"] == 8  # four passes per step
    assert set(a["kernels_ms"]) == set(b["kernels_ms"]) and a["kernels_ms"]["k1_lookup"] > 0


def test_gpu_reads_with_one_run_per_kmer(built, tmp_path):
    """a homopolymer read has as many minimizer runs as k-mers (all m-mers tie, the leftmost wins): consecutive such reads fill the
    lookup kernel's run queue beyond what waits in it, and here they MATCH (the index holds the k-mer A^31), so the heads of the
    pass in front are written while their runs wait. Mixed with ordinary reads, in one ticket and across tickets, 150 and 158 bases."""
    from oracle.pyoracle import OracleIndex
    rng = np.random.default_rng(31)
    k, n = 31, 12
    unitigs = [(0, "A" * k), (1, ("AC" * 16)[:k])]
    for sid in range(2, 8):
        unitigs.append((sid, "".join("ACGT"[c] for c in rng.integers(0, 4, size=220))))
    sets = [np.array([0, 3, 7]), np.array([1, 2]), np.arange(n), np.array([5]), np.arange(0, n, 2), np.arange(3, 9), np.array([0, 11]), np.arange(1, n)]
    base = str(tmp_path / "runs")
    with open(base + ".metadata.txt", "w") as f:
        f.write("k=%d\nnum_kmers=%d\nnum_colors=%d\nnum_unitigs=%d\nnum_color_sets=%d\n" %
                (k, sum(len(u) - k + 1 for _, u in unitigs), n, len(unitigs), len(sets)))
    with open(base + ".filenames.txt", "w") as f:
        f.write("".join("g%d.fa\n" % i for i in range(n)))
    with open(base + ".unitigs.fa", "w") as f:
        for sid, seq in unitigs:
            f.write("> color_set_id=%d\n%s\n" % (sid, seq))
    with open(base + ".color_sets.txt", "w") as f:
        for s_ in sets:
            f.write("size=%d %s\n" % (len(s_), " ".join(map(str, s_))))
    ix = fulgor_amd.Index(base, device=0)
    orc = OracleIndex.from_dump(base)
    long_u = [u for _, u in unitigs[2:]]
    reads = []
    for i in range(4000):
        kind = i % 8
        if kind in (0, 1, 2):
            reads.append("A" * (150 if i % 3 else 158))
        elif kind == 3:
            reads.append("T" * 150)
        elif kind == 4:
            reads.append(("AC" * 80)[:150])
        elif kind == 5:
            u = long_u[int(rng.integers(len(long_u)))]
            reads.append(u[int(rng.integers(0, 60)):][:150])
        elif kind == 6:
            reads.append("A" * 70 + long_u[0][:80])
        else:
            reads.append("C" * 150)
    reads = [r.encode() for r in reads]
    b, o = pack_reads(reads)
    for got, want in ((ix.fetch_color_set_ids_batch(b, o), orc.fetch_color_set_ids(b, o, threads=8)),
                      (ix.pseudoalign_full_intersection_batch(b, o), orc.full_intersection(b, o, threads=8)),
                      (ix.pseudoalign_threshold_union_batch(b, o, 0.7), orc.threshold_union(b, o, 0.7, threads=8))):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    sizes = np.diff(ix.pseudoalign_full_intersection_batch(b, o)[0].astype(np.int64))
    assert sizes[0] == 3 and sizes[3] == 3 and sizes[4] == 2 and sizes[7] == 0  # A^150, T^150 -> {0, 3, 7}; (AC)^75 -> {1, 2}; C^150 -> nothing
    ix.close()


@pytest.mark.parametrize("index_type,psize,csize", [(fulgor_amd.HYBRID, 0, 0), (fulgor_amd.DIFF, 4546, 16), (fulgor_amd.META, 160, 1), (fulgor_amd.META_DIFF, 160, 16)])
def test_gpu_matches_golden_at_4546_colours(s4546small, colour_stage, index_type, psize, csize):
    """golden vectors AT 4546 COLOURS from the dump-level first-principles oracle (oracle/dump_oracle.py: python sets over the
    dump text, nothing of the restatement): lookup, both algorithms, every codec, on the dense rows and on the codec's own kernels"""
    ix, _, _, fg, _ = s4546small
    if index_type != fulgor_amd.HYBRID:
        ix = fulgor_amd.Index(fg, device=0).convert(index_type, psize, csize)
    b, o = pack_reads(load_golden_reads("s4546small_reads.fa"))
    with stage(ix, colour_stage):
        fi = ix.pseudoalign_full_intersection_batch(b, o)
        tu = {tau: ix.pseudoalign_threshold_union_batch(b, o, tau) for tau in (0.8, 0.3)}
    assert csr_to_lists(*fi) == load_golden_tsv("s4546small_full_intersection.tsv.gz")
    for tau in (0.8, 0.3):
        assert csr_to_lists(*tu[tau]) == load_golden_tsv("s4546small_threshold_union_%s.tsv.gz" % tau)


# ---- round 5: the native worker loop (fgpu_pseudoalign_stream) ------------------------------------------------------------------
def _stream(ix, path, fmt, algo=fulgor_amd.FULL_INTERSECTION, tau=0.0, first_id=0, batch=0, workers=0, threads=3, **reader_kw):
    """one run of the native loop into a temporary file; returns (bytes, reads, mapped)"""
    import tempfile
    from fulgor_amd.reads import FastxReader
    rd = FastxReader(path, copy=False, threads=threads, **reader_kw)
    with tempfile.TemporaryFile() as out:
        n, mapped = ix.pseudoalign_stream(rd, out.fileno(), algo, tau, fmt, first_id, True, batch, workers)
        rd.close()
        out.seek(0)
        return out.read(), n, mapped


def test_stream_loop_equals_batch_calls_on_ragged_and_long_reads(s10_gpu, s10_oracle, tmp_path, monkeypatch):
    """fgpu_pseudoalign_stream against the host-buffer calls and the oracle: reads of 0 .. 60000 bases in one FASTA file (multi-line
    records; reads above 512 k-mers take the loop's segment path), ranges of 64 KB so that the file is many chunks, batches of 7 to 4000
    reads on 1 to 6 workers, the three output formats, full intersection and threshold union, read ids counted from an offset"""
    from oracle.kmer_oracle import read_fasta
    from oracle.pyoracle import parse_compressed
    from fulgor_amd.driver import Formatter
    monkeypatch.setenv("FULGOR_READER_RANGE_KB", "64")
    src = max(read_fasta(S10_GENOMES[5]), key=len)
    rng = np.random.default_rng(5)
    lens = [1054, 700, 300, 151, 31, 30, 0, 64, 5000, 60000, 1100] + [int(x) for x in rng.integers(0, 400, size=3000)] + [2078, 150, 150]
    reads = [src[(i * 977) % 3000000:(i * 977) % 3000000 + l] for i, l in enumerate(lens)]
    reads[20] = reads[20].replace(b"A", b"N", 2)
    fa = tmp_path / "ragged.fa"
    with open(fa, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">r%d some text\n" % i + b"".join(r[j:j + 80] + b"\n" for j in range(0, len(r), 80)) + (b"\n" if not r else b""))
    b, o = pack_reads(reads)
    fo, fc = s10_oracle.full_intersection(b, o)
    to, tc = s10_oracle.threshold_union(b, o, 0.7)
    nc = s10_gpu.num_colors()
    for batch, workers in ((7, 3), (500, 6), (4000, 1), (0, 0)):
        out, n, mapped = _stream(s10_gpu, str(fa), 0, batch=batch, workers=workers, first_id=1000)
        assert n == len(reads) and mapped == int((np.diff(fo.astype(np.int64)) > 0).sum())
        assert out == Formatter("ascii", nc).add(1000, fo, fc), (batch, workers)
    out, n, _ = _stream(s10_gpu, str(fa), 1, batch=900, workers=4)
    assert out == Formatter("binary", nc).add(0, fo, fc)
    out, n, mapped = _stream(s10_gpu, str(fa), 2, fulgor_amd.THRESHOLD_UNION, 0.7, batch=333, workers=5, first_id=17)
    ids, po, pc = parse_compressed(out)
    assert np.array_equal(ids, np.arange(17, 17 + len(reads), dtype=np.uint32)) and np.array_equal(po, to) and np.array_equal(pc, tc)
    assert mapped == int((np.diff(to.astype(np.int64)) > 0).sum())


def test_batches_that_hold_only_empty_reads(s10_gpu, tmp_path):
    """a batch whose reads have no bases at all (one empty record between two batch cuts of a ragged file): the lookup kernel's span of
    such a ticket is empty, and it used to read a kilobyte per round off a base buffer that may be a kilobyte long (a memory fault
    whenever the buffer ended a mapped block: once in a dozen runs of the ragged test above). Host-buffer calls and the streamed loop."""
    from fulgor_amd.driver import Formatter
    nc = s10_gpu.num_colors()
    for n in (1, 2, 5, 64, 65, 300):
        b, o = pack_reads([b""] * n)
        offs, cols = s10_gpu.pseudoalign_full_intersection_batch(b, o)
        assert len(cols) == 0 and np.array_equal(offs, np.zeros(n + 1, dtype=offs.dtype))
        offs, cols = s10_gpu.pseudoalign_threshold_union_batch(b, o, 0.5)
        assert len(cols) == 0 and len(offs) == n + 1
    fa = tmp_path / "empties.fa"
    fa.write_bytes(b"".join(b">e%d\n\n" % i for i in range(40)))
    for batch, workers in ((1, 3), (7, 2), (0, 0)):
        for _ in range(5):
            out, n, mapped = _stream(s10_gpu, str(fa), 0, batch=batch, workers=workers)
            assert (n, mapped) == (40, 0) and out == Formatter("ascii", nc).add(0, np.zeros(41, dtype=np.uint64), np.zeros(0, dtype=np.uint32))


@pytest.mark.parametrize("which", ["s10", "s4546small", "s4546small-blocks", "s4546small-metadiff-codec", "s10-diff-codec"])
def test_stream_loop_fuzz_against_the_batch_calls(which, s10_gpu, s10_fgidx, s10_oracle, s4546small, tmp_path, monkeypatch):
    """seeded fuzz of the worker loop: FASTA (single- or multi-line) and four-line FASTQ files of reads of every awkward length (0, k - 1, k,
    129 .. 600 k-mers, one of 20000 bases, runs of empty records, N bases), ranges of 4 .. 256 KB, FASTQ pieces of 1 .. 64 KB, batches of 1 .. 5000
    reads on 1 .. 7 workers, both algorithms, ascii and binary records: byte-identical to the formatter over the host-buffer calls
    (which the tests above hold against the oracle, and every fourth file here again: the codec of an index changes nothing about
    its answers, so the hybrid oracle serves the converted indexes too); compressed records parse back to the same lists"""
    from oracle.kmer_oracle import read_fasta
    from oracle.pyoracle import parse_compressed
    from fulgor_amd.driver import Formatter
    # (the 4546-colour index: results of hundreds of colours, every record kind of the compressed format, rows of 144 words)
    # (-blocks: the hybrid lists on their own packed-block kernels; -codec: a converted index on the codec's kernels, no dense rows)
    own = None
    if which == "s10":
        ix = s10_gpu
    elif which == "s10-diff-codec":
        ix = own = fulgor_amd.Index(s10_fgidx, device=0).convert(fulgor_amd.DIFF, 10, 4)
    elif which == "s4546small-metadiff-codec":
        ix = own = fulgor_amd.Index(s4546small[3], device=0).convert(fulgor_amd.META_DIFF, 160, 16)
    elif which == "s4546small-blocks":
        ix = own = fulgor_amd.Index(s4546small[3], device=0)
    else:
        ix = s4546small[0]
    if own is not None:
        own.tune(dense_rows=False)
    s10_gpu = ix
    src = max(read_fasta(S10_GENOMES[3 if which.startswith("s10") else 0]), key=len)
    rng = np.random.default_rng(20250930)
    nc = s10_gpu.num_colors()
    special = [0, 0, 0, 30, 31, 32, 150, 158, 159, 160, 286, 287, 542, 543, 630, 20000]
    scale = int(os.environ.get("FULGOR_TEST_FUZZ_SCALE", "1"))  # (a one-off soak: FULGOR_TEST_FUZZ_SCALE=10)
    for trial in range(scale * (160 if which == "s10" else (120 if which == "s4546small" else 40))):
        n = int(rng.integers(1, 2500))
        lens = [int(x) for x in rng.integers(0, int(rng.choice([40, 200, 400, 700])), size=n)]
        for _ in range(int(rng.integers(0, 12))):
            lens[int(rng.integers(0, n))] = int(rng.choice(special))
        if trial % 3 == 0:  # a run of empty records long enough to fill batches of their own
            at = int(rng.integers(0, n))
            lens[at:at + 70] = [0] * len(lens[at:at + 70])
        reads = []
        for i, l in enumerate(lens):
            st = int(rng.integers(0, len(src) - 20001))
            r = bytearray(src[st:st + l])
            if l and i % 97 == 5:
                r[int(rng.integers(0, l))] = ord("N")
            reads.append(bytes(r))
        fastq = trial % 2 == 1
        path = tmp_path / ("fuzz%d.%s" % (trial, "fq" if fastq else "fa"))
        with open(path, "wb") as f:
            for i, r in enumerate(reads):
                if fastq:
                    f.write(b"@q%d\n%s\n+\n%s\n" % (i, r, b"@" * len(r)))
                else:
                    w = int(rng.choice([60, 80, 100000]))
                    f.write(b">s%d t\n" % i + b"".join(r[j:j + w] + b"\n" for j in range(0, len(r), w)) + (b"\n" if not r else b""))
        if trial % 5 == 2:  # the same text as an ordinary gzip file (inflated whole and parsed in ranges, or streamed: FULGOR_GZIP_STREAM)
            import gzip
            gz = str(path) + ".gz"
            with open(path, "rb") as f, open(gz, "wb") as g_:
                g_.write(gzip.compress(f.read(), 1))
            path = gz
            if trial % 10 == 2:
                monkeypatch.setenv("FULGOR_GZIP_STREAM", "1")
            else:
                monkeypatch.delenv("FULGOR_GZIP_STREAM", raising=False)
        monkeypatch.setenv("FULGOR_READER_RANGE_KB", str(int(rng.choice([4, 16, 64, 256]))))
        monkeypatch.setenv("FULGOR_READER_PIECE_KB", str(int(rng.choice([1, 3, 64]))))
        b, o = pack_reads(reads)
        algo, tau = (fulgor_amd.THRESHOLD_UNION, float(rng.choice([0.3, 0.8, 1.0]))) if trial % 4 >= 2 else (fulgor_amd.FULL_INTERSECTION, 0.0)
        if algo == fulgor_amd.THRESHOLD_UNION:
            eo, ec = s10_gpu.pseudoalign_threshold_union_batch(b, o, tau)
        else:
            eo, ec = s10_gpu.pseudoalign_full_intersection_batch(b, o)
        if trial % 4 == 0:
            orc = s10_oracle if which.startswith("s10") else s4546small[1]
            oo, oc = orc.threshold_union(b, o, tau) if algo == fulgor_amd.THRESHOLD_UNION else orc.full_intersection(b, o)
            assert np.array_equal(eo, oo) and np.array_equal(ec, oc), (which, trial)
        first_id = int(rng.integers(0, 1000))
        if trial % 7 == 3 and not str(path).endswith(".gz"):
            # as the ranks of `--gpus N` do: byte ranges of the file cut anywhere, read ids continued from part to part
            size = os.path.getsize(path)
            cuts = sorted({0, size} | {int(x) for x in rng.integers(0, size + 1, size=int(rng.integers(1, 4)))})
            joined, at = b"", 0
            for a_, b_ in zip(cuts[:-1], cuts[1:]):
                out, got, _ = _stream(s10_gpu, str(path), 0, algo, tau, first_id=first_id + at, batch=int(rng.choice([7, 333, 5000])),
                                      workers=int(rng.integers(1, 6)), begin=a_, end=b_)
                joined += out
                at += got
            assert at == n and joined == Formatter("ascii", nc).add(first_id, eo, ec), (which, trial, cuts)
        for rep in range(3):
            batch, workers, fmt = int(rng.choice([1, 7, 64, 333, 5000])), int(rng.integers(1, 8)), rep
            out, got, mapped = _stream(s10_gpu, str(path), fmt, algo, tau, first_id=first_id, batch=batch, workers=workers, threads=int(rng.integers(1, 5)))
            assert got == n and mapped == int((np.diff(eo.astype(np.int64)) > 0).sum()), (trial, rep)
            if fmt == 2:
                ids, po, pc = parse_compressed(out)
                assert np.array_equal(ids, np.arange(first_id, first_id + n, dtype=np.uint32)) and np.array_equal(po, eo) and np.array_equal(pc, ec), (trial, batch, workers)
            else:
                assert out == Formatter(("ascii", "binary")[fmt], nc).add(first_id, eo, ec), (trial, batch, workers, fmt)


def test_stream_loop_on_empty_wrapped_and_broken_files(s10_gpu, s10_oracle, tmp_path):
    """an empty query file gives the header and no records; a FASTQ file that turns into wrapped lines behind a four-line head offers no record
    boundaries in its tail, which then falls to one range and the full grammar: same records as the oracle's; a gzip file with a flipped byte
    in the middle ends the run with the reader's error — no hang, no partial success — and the loop works again afterwards"""
    import gzip
    from fulgor_amd.driver import Formatter
    empty = tmp_path / "empty.fq"
    empty.write_bytes(b"")
    out, n, mapped = _stream(s10_gpu, str(empty), 2)
    assert (n, mapped) == (0, 0) and out == Formatter("compressed", s10_gpu.num_colors()).header
    out, n, mapped = _stream(s10_gpu, str(empty), 0)
    assert (n, mapped, out) == (0, 0, b"")
    seqs = [r for r in load_golden_reads()[:3000]]
    good = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)) for i, r in enumerate(seqs))
    tail = b"".join(b"@w%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:70], r[70:], b"@" * 70, b"I" * (len(r) - 70)) for i, r in enumerate(seqs[:1500]) if len(r) > 70)
    p = tmp_path / "wrapped_tail.fq"
    p.write_bytes(good + tail)
    b, o = pack_reads(seqs + [r for r in seqs[:1500] if len(r) > 70])
    fo, fc = s10_oracle.full_intersection(b, o)
    os.environ["FULGOR_READER_RANGE_KB"] = "64"
    try:
        out, n, mapped = _stream(s10_gpu, str(p), 0, batch=1000, workers=4)
        assert n == len(o) - 1 and out == Formatter("ascii", s10_gpu.num_colors()).add(0, fo, fc)
        z = bytearray(gzip.compress(good * 4, 6))
        z[len(z) // 2] ^= 0x55
        pz = tmp_path / "broken.fq.gz"
        pz.write_bytes(bytes(z))
        with pytest.raises(RuntimeError):
            _stream(s10_gpu, str(pz), 0, batch=1000, workers=4)
        out2, n2, _ = _stream(s10_gpu, str(p), 0, batch=1000, workers=4)
        assert n2 == n and out2 == out
    finally:
        del os.environ["FULGOR_READER_RANGE_KB"]


@pytest.mark.parametrize("which", ["s10", "s4546small"])
def test_kmer_tools_fuzz_against_the_oracle(which, s10_gpu, s10_oracle, s4546small):
    """kmer-conservation and kmer-matches (src/kmer_conservation.cpp:7-54, src/kmer_matches.cpp:7-30) on batches of awkward reads — empty, shorter
    than k, exactly k, 129 / 257 / 513 k-mers, N bases, lower case, one of 9000 bases, runs of empty reads — against the oracle read by read:
    per-k-mer colour-set ids as conservation triples, positive-k-mer flags, per-colour match counts"""
    from oracle.kmer_oracle import read_fasta
    from fulgor_amd.index import conservation_triples
    ix, orc = (s10_gpu, s10_oracle) if which == "s10" else (s4546small[0], s4546small[1])
    src = max(read_fasta(S10_GENOMES[2 if which == "s10" else 0]), key=len)
    rng = np.random.default_rng(777)
    special = [0, 0, 1, 30, 31, 32, 158, 159, 160, 286, 287, 288, 542, 543, 544, 9000]
    # the native line emitters of the two tools (fgpu_kmer_emitter_*: what the command line writes), fed the same batches one after the
    # other: their state — what a record shorter than k repeats (src/kmer_matches.cpp:11) — carries over from batch to batch
    from fulgor_amd.index import KmerEmitter
    em_c, em_m = KmerEmitter(ix, 0), KmerEmitter(ix, 1)
    prev = (np.zeros(0, dtype=np.uint8), np.zeros(ix.num_colors(), dtype=np.uint32))
    for trial in range(60 * int(os.environ.get("FULGOR_TEST_FUZZ_SCALE", "1"))):
        n = int(rng.integers(1, 400))
        lens = [int(x) for x in rng.integers(0, int(rng.choice([50, 200, 600])), size=n)]
        for _ in range(int(rng.integers(1, 10))):
            lens[int(rng.integers(0, n))] = int(rng.choice(special))
        if trial % 3 == 1:
            lens[:70] = [0] * len(lens[:70])
        reads = []
        for i, l in enumerate(lens):
            st = int(rng.integers(0, len(src) - 9001))
            r = bytearray(src[st:st + l])
            if l and i % 11 == 3:
                r[int(rng.integers(0, l))] = ord("N")
            if l and i % 7 == 2:
                r = bytearray(bytes(r).lower())
            reads.append(bytes(r))
        b, o = pack_reads(reads)
        ko, ki = ix.kmer_color_set_ids_batch(b, o)
        mo, pos, counts = ix.kmer_matches_batch(b, o)
        assert np.array_equal(ko, mo) and len(ko) == n + 1
        for j, r in enumerate(reads):
            a, e = int(ko[j]), int(ko[j + 1])
            assert e - a == max(0, len(r) - 30), (trial, j)
            assert conservation_triples(ki[a:e]) == orc.kmer_conservation(r), (which, trial, j, len(r))
            if j % 5 == 0 or len(r) in special:
                opos, ocnt = orc.kmer_matches(r)
                assert np.array_equal(pos[a:e], opos) and np.array_equal(counts[j], ocnt), (which, trial, j, len(r))
        # the emitters' lines for this batch against lines made of the answers above (those were checked against the oracle)
        names = [("rec%d_%d" % (trial, j)).encode() for j in range(n)]
        nb, no = pack_reads(names)
        want_c, want_m = [], []
        for j, r in enumerate(reads):
            a, e = int(ko[j]), int(ko[j + 1])
            tr = conservation_triples(ki[a:e])
            want_c.append(names[j] + b"\t%d" % len(tr) + b"".join(b"\t(%d %d %d)" % t for t in tr) + b"\n")
            if len(r) >= 31:
                prev = (pos[a:e], counts[j])
            want_m.append(names[j] + b"\t%d" % len(prev[0]) + b"".join(b"\t%d" % x for x in prev[0]) + b"".join(b"\t%d" % x for x in prev[1]) + b"\n")
        assert bytes(em_c.add(b, o, nb, no, n)) == b"".join(want_c), (which, trial)
        if which == "s10" or trial % 6 == 0:  # (4546 counts per line: a sixth of the batches)
            assert bytes(em_m.add(b, o, nb, no, n)) == b"".join(want_m), (which, trial)
        else:
            prev_skip = [j for j, r in enumerate(reads) if len(r) >= 31]
            if prev_skip:  # the emitter did not see this batch: bring its state along through a batch of that one record
                j = prev_skip[-1]
                b1, o1 = pack_reads([reads[j]])
                n1, no1 = pack_reads([names[j]])
                em_m.add(b1, o1, n1, no1, 1)
    em_c.close()
    em_m.close()


def test_stream_loop_reports_an_output_that_cannot_be_written(s10_gpu, tmp_path):
    """the records cannot be written (a descriptor opened for reading; /dev/full): the call fails with the system's message instead of
    hanging or dropping records silently (tools/pseudoalign.cpp lets the stream's failure surface the same way), the workers and the
    reader's threads wind down, and the next run on the same index is complete"""
    from fulgor_amd.reads import FastxReader
    fq = tmp_path / "w.fq"
    fq.write_bytes(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, b"ACGT" * 40, b"I" * 160) for i in range(30000)))
    for target, flags in ((str(fq), os.O_RDONLY), ("/dev/full", os.O_WRONLY)):
        if not os.path.exists(target):
            continue
        rd = FastxReader(str(fq), copy=False, threads=3)
        fd = os.open(target, flags)
        try:
            with pytest.raises(RuntimeError, match="cannot write the output"):
                s10_gpu.pseudoalign_stream(rd, fd, fulgor_amd.FULL_INTERSECTION, 0.0, 0, 0, True, 1000, 4)
        finally:
            os.close(fd)
            rd.close()
    out, n, _ = _stream(s10_gpu, str(fq), 0, batch=1000, workers=4)
    assert n == 30000 and out.count(b"\n") == 30000


def test_host_buffer_calls_from_several_threads(s10_gpu, seeded_reads):
    """fgpu_full_intersection / fgpu_threshold_union as a reference worker pool calls them: one call per chunk of reads from several
    threads at once (the binding releases the GIL inside the call). The index hands every call a result of its own — one a finished call
    left behind, or a new one — and the outputs come from (and go back to) the pool of pinned slabs: every chunk's lists equal the ones
    a single thread gets, chunk sizes from one read to a few thousand, six rounds over eight threads."""
    import threading
    ix = s10_gpu
    b, o = seeded_reads
    rng = np.random.default_rng(17)
    cuts = [0]
    while cuts[-1] < 40000:
        cuts.append(min(40000, cuts[-1] + int(rng.integers(1, 4000))))
    chunks = []
    for a, e in zip(cuts[:-1], cuts[1:]):
        lo, hi = int(o[a]), int(o[e])
        chunks.append((np.ascontiguousarray(b[lo:hi]), np.ascontiguousarray(o[a:e + 1] - o[a])))
    want = [(ix.pseudoalign_full_intersection_batch(cb, co), ix.pseudoalign_threshold_union_batch(cb, co, 0.7)) for cb, co in chunks]
    bad = []

    def work(k):
        for _ in range(6):
            for i in range(k, len(chunks), 8):
                cb, co = chunks[i]
                fi = ix.pseudoalign_full_intersection_batch(cb, co)
                tu = ix.pseudoalign_threshold_union_batch(cb, co, 0.7)
                if not (np.array_equal(fi[0], want[i][0][0]) and np.array_equal(fi[1], want[i][0][1]) and
                        np.array_equal(tu[0], want[i][1][0]) and np.array_equal(tu[1], want[i][1][1])):
                    bad.append(i)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, sorted(set(bad))[:10]


def test_two_streamed_runs_at_once_on_one_index(s10_gpu, tmp_path):
    """two callers stream two different files through the same index at the same time (a server that answers two requests): each gets
    its own workers from the index's cache, the pinned slabs and the copy engines are shared; both outputs equal what the runs give
    one after the other, several times over"""
    import threading
    from oracle.kmer_oracle import read_fasta
    src = max(read_fasta(S10_GENOMES[1]), key=len)
    files = []
    for k_, n in ((0, 40000), (1, 25000)):
        p = tmp_path / ("c%d.fq" % k_)
        with open(p, "wb") as f:
            for i in range(n):
                st = (i * 7919 + k_ * 1000003) % (len(src) - 200)
                r = src[st:st + 100 + (i % 60)]
                f.write(b"@x%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
        files.append((str(p), n))
    alone = [_stream(s10_gpu, path, 0, batch=3000, workers=3) for path, _ in files]
    for rep in range(4):
        got = [None, None]

        def run(k_):
            got[k_] = _stream(s10_gpu, files[k_][0], 0, batch=3000 - 500 * rep, workers=2 + rep % 3)
        ts = [threading.Thread(target=run, args=(k_,)) for k_ in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for k_ in range(2):
            assert got[k_] is not None and got[k_][1] == files[k_][1] and got[k_][0] == alone[k_][0], (rep, k_)


def test_dictionary_of_more_than_2_pow_26_buckets(built, tmp_path):
    """round-5 review, weak item 11: a collection whose dictionary needs more than 2^26 buckets (about 120 M distinct 31-mers) was refused at
    load — a (bucket, lane) pair of the lookup kernel's ring had to fit 32 bits. Such tables now run the kernel's WIDE instantiations
    (the ring keeps the lane in a word of its own; up to 2^31 buckets). (1) 140 random unitigs of a million bases: 140 M k-mers, 47 M
    super-k-mer records, 76 M hashed buckets, a 5 GB table built on the device; the self check compares it byte for byte with the host
    builder's and walks every k-mer of every 28th unitig through it on both strands; reads cut out of the unitigs (both strands) fetch
    their unitig's colour-set id and intersect to its colours, reads of random bases fetch nothing. (2) The WIDE instantiations forced
    onto the ordinary indexes (FULGOR_DICT_WIDE=1): the parity tests of the lookup run again in a process of their own."""
    import subprocess
    rng = np.random.default_rng(26)
    nu, ulen, ncol = 140, 1_000_000, 4
    sets = [[0], [1, 3], [0, 1, 2, 3]]
    base = str(tmp_path / "wide")
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    units = [alpha[rng.integers(0, 4, size=ulen, dtype=np.uint8)] for _ in range(nu)]
    csid = sorted(u % 3 for u in range(nu))  # (a dump lists its unitigs by colour-set id)
    with open(base + ".unitigs.fa", "wb") as f:
        for u in range(nu):
            f.write(b"> color_set_id=%d\n" % csid[u])
            f.write(units[u].tobytes())
            f.write(b"\n")
    open(base + ".color_sets.txt", "w").write("".join("size=%d %s\n" % (len(x), " ".join(map(str, x))) for x in sets))
    open(base + ".filenames.txt", "w").write("".join("g%d.fa\n" % c for c in range(ncol)))
    open(base + ".metadata.txt", "w").write("k=31\nnum_kmers=%d\nnum_colors=%d\nnum_unitigs=%d\nnum_color_sets=%d\n" % (nu * (ulen - 30), ncol, nu, len(sets)))
    ix = fulgor_amd.Index(base, device=0)
    assert ix.num_kmers() == nu * (ulen - 30)
    ix.selfcheck(unitig_stride=28)
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    reads, want_ids = [], []
    for i in range(20000):
        u = int(rng.integers(0, nu))
        st = int(rng.integers(0, ulen - 150))
        r = units[u][st:st + 150]
        if i % 2:
            r = comp[r[::-1]]
        reads.append(r.tobytes())
        want_ids.append(csid[u])
    for i in range(500):
        reads.append(alpha[rng.integers(0, 4, size=150, dtype=np.uint8)].tobytes())
        want_ids.append(None)
    b, o = pack_reads(reads)
    io_, ii = ix.fetch_color_set_ids_batch(b, o)
    io_ = io_.astype(np.int64)
    fo, fc = ix.pseudoalign_full_intersection_batch(b, o)
    fo = fo.astype(np.int64)
    for i, w in enumerate(want_ids):
        got = ii[io_[i]:io_[i + 1]].tolist()
        assert got == ([] if w is None else [w]), (i, got, w)
        assert fc[fo[i]:fo[i + 1]].tolist() == ([] if w is None else sets[w]), i
    ko, ki = ix.kmer_color_set_ids_batch(b[:150 * 200], o[:201])
    assert (ki.reshape(200, 120) == np.array(want_ids[:200], dtype=np.uint32)[:, None]).all()  # every k-mer of a read cut out of a unitig is found
    ix.close()
    pick = ("fetch_color_set_ids_equals_oracle or full_intersection_matches_golden or reads_of_any_length or reads_up_to_512 or one_run_per_kmer or "
            "fuzz_dirty_ragged or matches_golden_at_4546 or other_kmer_lengths")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k", pick,
                        "-p", "no:cacheprovider"], env=dict(os.environ, FULGOR_DICT_WIDE="1"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_no_kernel_leaves_its_buffers_under_the_guard_allocator(built):
    """FULGOR_GUARD_ALLOC=1: every device buffer is exactly as long as asked for and is followed by unmapped addresses, so a kernel
    that reads or writes past a buffer faults at once (the lookup kernel's empty-ticket read of round 5 needed a buffer that ended a
    mapped block to show: once in a dozen runs). The tests of awkward inputs run again in that mode, in a process of their own."""
    import subprocess
    pick = ("fuzz_against_the_batch or kmer_tools_fuzz or only_empty_reads or ragged_and_long or reads_of_any_length or other_kmer_lengths or "
            "empty_wrapped_and_broken or device_formatters_are_byte_identical or compressed_formatter_parses_back or "
            "device_side_deduplication or materialised_only_on_demand")  # (round 6: the grouping kernels, the checksum kernels; every open builds the dictionary table on the device)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k", pick,
                        "-p", "no:cacheprovider"], env=dict(os.environ, FULGOR_GUARD_ALLOC="1"), capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and " passed" in r.stdout and "Memory access fault" not in tail, tail
