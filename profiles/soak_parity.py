"""One-off soak of the HIP path against the oracle on the bench index, far beyond the sizes the test suite affords: full
intersection, threshold union, fetched ids on millions of seeded reads, plus dirty reads (random N, random lengths).
usage (GPU box): python profiles/soak_parity.py [reads in millions, default 2]"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import ReadGenerator
from oracle.pyoracle import OracleIndex
M = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
ix = fulgor_amd.Index(fg, device=0)
orc = OracleIndex.from_export(ix.export())
gen = ReadGenerator(g, raw_sequences=extra)
T = min(64, os.cpu_count() or 8)
bad = 0


def check(tag, a, b):
    global bad
    ok = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))
    bad += not ok
    print("%-46s %s" % (tag, "equal" if ok else "DIFFERENT"), flush=True)


n = int(M * 1e6)
for seed, first in ((7, 50_000_000), (8, 90_000_000)):
    b, o = gen.generate(first, n, 150, seed)
    t0 = time.time()
    check("full intersection, %d reads, seed %d" % (n, seed), ix.pseudoalign_full_intersection_batch(b, o), orc.full_intersection(b, o, threads=T))
    check("fetched ids, %d reads, seed %d" % (n, seed), ix.fetch_color_set_ids_batch(b, o), orc.fetch_color_set_ids(b, o, threads=T))
    print("  (%.0f s)" % (time.time() - t0), flush=True)
b, o = gen.generate(123_000_000, n // 2, 150, 9)
for tau in (0.8, 0.25, 0.5):  # (k3r_union: five planes of deficit counters, byte counters, six planes)
    check("threshold union tau %.2f, %d reads" % (tau, n // 2), ix.pseudoalign_threshold_union_batch(b, o, tau), orc.threshold_union(b, o, tau, threads=T))
# longer reads: the lookup kernel's instantiations for up to 192 / 256 / 384 / 512 k-mers, the threshold union's six / seven planes and 16-bit counters
for length, tau in ((200, 0.8), (300, 0.8), (300, 0.5), (540, 0.8)):
    bl, ol = gen.generate(200_000_000 + length, 300_000, length, 12)
    check("reads of %d bases, full intersection" % length, ix.pseudoalign_full_intersection_batch(bl, ol), orc.full_intersection(bl, ol, threads=T))
    check("reads of %d bases, threshold union %.1f" % (length, tau), ix.pseudoalign_threshold_union_batch(bl, ol, tau), orc.threshold_union(bl, ol, tau, threads=T))
# dirty, ragged reads: random lengths 0..400, 3 % N, lower case
rng = np.random.default_rng(5)
b, o = gen.generate(7_000_000, 200_000, 400, 10)
b = np.array(b)
lens = rng.integers(0, 401, size=200_000)
keep = np.concatenate([np.arange(int(o[i]), int(o[i]) + int(lens[i])) for i in range(200_000)])
b2 = b[keep]
o2 = np.concatenate(([0], np.cumsum(lens))).astype(np.uint64)
mask = rng.random(len(b2)) < 0.03
b2[mask] = ord("N")
low = rng.random(len(b2)) < 0.2
b2[low] = np.char.lower(b2[low].view("S1")).view(np.uint8)
check("dirty ragged reads, full intersection", ix.pseudoalign_full_intersection_batch(b2, o2), orc.full_intersection(b2, o2, threads=T))
check("dirty ragged reads, threshold union 0.5", ix.pseudoalign_threshold_union_batch(b2, o2, 0.5), orc.threshold_union(b2, o2, 0.5, threads=T))
# per-k-mer output (kmer-conservation): 20 k reads, one oracle call each
from fulgor_amd.index import conservation_triples
b, o = gen.generate(31_000_000, 20000, 150, 11)
ko, ki = ix.kmer_color_set_ids_batch(b, o)
bb = bytes(np.asarray(b))
same = all(conservation_triples(ki[int(ko[j]):int(ko[j + 1])]) == orc.kmer_conservation(bb[int(o[j]):int(o[j + 1])]) for j in range(20000))
bad += not same
print("%-46s %s" % ("k-mer conservation triples, 20000 reads", "equal" if same else "DIFFERENT"), flush=True)
# (round 5) the streamed worker loop (fgpu_pseudoalign_stream: parser threads -> pinned chunks -> five workers -> compressed records in file
# order) on a FASTQ file of the same kind of reads: its records, parsed back by the oracle's reader, against the oracle's results
from oracle.pyoracle import parse_compressed
from fulgor_amd.reads import FastxReader
ns = n // 2
b, o = gen.generate(300_000_000, ns, 150, 13)
path = "/dev/shm/soak_%d.fq" % os.getpid()
rec = np.empty((ns, 316), dtype=np.uint8)
idn = np.arange(ns, dtype=np.int64)
rec[:, 0], rec[:, 1], rec[:, 11] = ord("@"), ord("r"), ord("\n")
for d in range(9):
    rec[:, 2 + d] = ord("0") + (idn // 10 ** (8 - d)) % 10
rec[:, 12:162] = np.asarray(b).reshape(ns, 150)
rec[:, 162:165] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 165:-1] = ord("I")
rec[:, -1] = ord("\n")
rec.tofile(path)
del rec
try:
    for algo, tau, want in ((0, 0.0, orc.full_intersection(b, o, threads=T)), (1, 0.8, orc.threshold_union(b, o, 0.8, threads=T))):
        out = path + ".out"
        rd = FastxReader(path, copy=False)
        fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
        got, mapped = ix.pseudoalign_stream(rd, fd, algo, tau, 2, 0, True, 0, 0)
        os.close(fd)
        rd.close()
        ids, po, pc = parse_compressed(open(out, "rb").read())
        os.remove(out)
        ok_ids = got == ns and np.array_equal(ids, np.arange(ns, dtype=np.uint32)) and mapped == int((np.diff(want[0].astype(np.int64)) > 0).sum())
        check("streamed loop, %s, %d reads (ids and counters %s)" % ("threshold union 0.8" if algo else "full intersection", ns, "ok" if ok_ids else "WRONG"), (po, pc), want)
        bad += not ok_ids
finally:
    os.remove(path)
# the other codecs against the hybrid result of the same reads (both HIP; the codecs meet the oracle on the small index)
b, o = gen.generate(200_000_000, n // 2, 150, 12)
ref_fi = ix.pseudoalign_full_intersection_batch(b, o)
ref_tu = ix.pseudoalign_threshold_union_batch(b, o, 0.8)
# (round 3) the reference results above came from the dense rows (k2r_intersect / k3r_union); the packed-block kernels of the
# hybrid index and the codec kernels must give the same
ix.tune(dense_rows=False)
check("hybrid on packed blocks (k2a), full intersection, %d reads" % (n // 2), ix.pseudoalign_full_intersection_batch(b, o), ref_fi)
check("hybrid on packed blocks (k3a), threshold union 0.8", ix.pseudoalign_threshold_union_batch(b, o, 0.8), ref_tu)
for t, name in ((3, "meta-differential"), (1, "differential"), (2, "meta")):
    ix.convert(t, 128, 16)
    check("%s, full intersection, %d reads" % (name, n // 2), ix.pseudoalign_full_intersection_batch(b, o), ref_fi)
    check("%s, threshold union 0.8, %d reads" % (name, n // 2), ix.pseudoalign_threshold_union_batch(b, o, 0.8), ref_tu)
print("SOAK", "FAILED" if bad else "PASSED")
sys.exit(1 if bad else 0)
