"""Generates the committed golden vectors ABOVE 64 COLOURS (run once, in the build container).

Inputs : the seeded 256-genome collection of synth_c256.py, k = 31
Reads  : synth_c256.reads(): 400 seeded 150-base reads + 8 edge cases
Outputs: c256_reads.fa, c256_full_intersection.tsv, c256_threshold_union_0.8.tsv, c256_threshold_union_0.3.tsv in the
         reference's ascii output format "<id>\t<count>[\t<colour>...]" (src/ps_utils.cpp:55-71)
Oracle : oracle/kmer_oracle.py — per-k-mer colour masks computed directly from the genomes; shares no code with the
         engine or with oracle/fulgor_oracle.hpp."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import synth_c256  # noqa: E402
from oracle.kmer_oracle import KmerOracle  # noqa: E402


def main():
    gen = synth_c256.genomes()
    reads = synth_c256.reads(gen)
    with open(os.path.join(HERE, "c256_reads.fa"), "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">r%d\n%s\n" % (i, r))
    orc = KmerOracle([[g] for g in gen], 31)

    def dump(name, fn):
        with open(os.path.join(HERE, name), "w") as f:
            for i, r in enumerate(reads):
                cols = fn(r)
                f.write("\t".join([str(i), str(len(cols))] + [str(c) for c in cols]) + "\n")

    dump("c256_full_intersection.tsv", orc.full_intersection)
    for tau in (0.8, 0.3):
        dump("c256_threshold_union_%s.tsv" % tau, lambda r, t=tau: orc.threshold_union(r, t))
    print("distinct canonical 31-mers:", len(orc.keys))


if __name__ == "__main__":
    main()
