"""Generates the committed golden vectors for the pseudoalignment path (run once, in the build
container; the outputs next to this script are what the tests read).

Inputs : tests/data/salmonella_10/*.fasta.gz  (the reference's own test_data/salmonella_10, colour id =
         position in the sorted filename list), k = 31
Reads  : 1000 synthetic 150 bp reads (seed 42, fulgor_amd/csrc/tools/readgen.cpp) + hand-made edge cases
Outputs: s10_reads.fa, s10_kmer_level.tsv (positive flags / per-colour counts / equal-mask runs),
         s10_full_intersection.tsv, s10_threshold_union_0.8.tsv, s10_threshold_union_1.0.tsv,
         s10_threshold_union_0.01.tsv in the reference's ascii output format "<id>\t<count>[\t<colour>...]"
         (src/ps_utils.cpp:55-71)
Oracle : oracle/kmer_oracle.py — per-k-mer colour masks computed directly from the genomes; shares no
         code with the engine or with oracle/fulgor_oracle.hpp.
"""
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from fulgor_amd.reads import ReadGenerator  # noqa: E402
from oracle.kmer_oracle import KmerOracle, read_fasta  # noqa: E402


def main():
    genomes = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
    assert len(genomes) == 10
    gen = ReadGenerator(genomes)
    bases, offs = gen.generate(0, 1000, 150, 42)
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(1000)]
    # edge cases: shorter than k, exactly k, N inside, empty, lower case, longer/ragged,
    # one window only valid, poly-A
    src = max(read_fasta(genomes[3]), key=len)  # longest contig of genome 3
    g = src[5000:5400]
    assert len(g) == 400 and set(g) <= set(b"ACGT")
    reads += [
        g[:30],                                  # 1000: len < k  -> empty result
        g[:31],                                  # 1001: exactly one k-mer
        g[:70] + b"N" + g[71:150],               # 1002: N in the middle
        b"",                                     # 1003: empty read
        g[:150].lower(),                         # 1004: lower case
        g[:400],                                 # 1005: 400 bp (370 k-mers)
        b"N" * 40 + g[100:131] + b"N" * 40,      # 1006: a single valid window
        b"A" * 150,                              # 1007: homopolymer
        g[:150][::-1],                           # 1008: reversed (not complemented)
        g[200:231] + b"ACGT" * 10,               # 1009: one genomic k-mer then junk
    ]
    with open(os.path.join(HERE, "s10_reads.fa"), "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">r%d\n%s\n" % (i, r))
    orc = KmerOracle(genomes, 31)

    def dump(name, fn):
        with open(os.path.join(HERE, name), "w") as f:
            for i, r in enumerate(reads):
                cols = fn(r)
                f.write("\t".join([str(i), str(len(cols))] + [str(c) for c in cols]) + "\n")

    dump("s10_full_intersection.tsv", orc.full_intersection)
    for tau in (0.8, 1.0, 0.01):
        dump("s10_threshold_union_%s.tsv" % tau, lambda r, t=tau: orc.threshold_union(r, t))
    # k-mer level golden vectors (kmer_matches / kmer_conservation): for a subset of reads, the positive flags,
    # the per-colour counts and the runs of consecutive positive k-mers with the same colour mask
    import numpy as np
    from oracle.kmer_oracle import canonical_kmers
    sel = list(range(0, 40)) + list(range(1000, 1010))
    with open(os.path.join(HERE, "s10_kmer_level.tsv"), "w") as f:
        for i in sel:
            km, ok = canonical_kmers(reads[i], 31)
            if len(km) == 0:
                f.write("%d\t\t\t\n" % i)
                continue
            idx = np.searchsorted(orc.keys, km)
            idx[idx >= len(orc.keys)] = 0
            hit = ok & (orc.keys[idx] == km)
            masks = np.where(hit, orc.masks[idx, 0], np.uint64(0))
            counts = [int(((masks >> np.uint64(c)) & np.uint64(1)).sum()) for c in range(orc.n)]
            runs, p = [], 0
            while p < len(masks):
                if not hit[p]:
                    p += 1
                    continue
                q = p
                while q < len(masks) and hit[q] and masks[q] == masks[p]:
                    q += 1
                runs.append("%d:%d:%d" % (p, q - p, int(masks[p])))
                p = q
            f.write("%d\t%s\t%s\t%s\n" % (i, "".join("1" if h else "0" for h in hit), ",".join(map(str, counts)), ",".join(runs)))
    print("distinct canonical 31-mers:", len(orc.keys))


if __name__ == "__main__":
    main()
