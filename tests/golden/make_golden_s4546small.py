"""Generates the committed golden vectors AT 4546 COLOURS (run once, in the build container).

Inputs : the small synthetic 4546-colour test index (fulgor_amd.synth.ensure_s4546_small: seeded, rebuilt identically
         everywhere) written out as the reference's dump files by fgpu_dump: data/s4546small_dump.*
Reads  : 96 seeded 150-base reads of the test generator (seed 4546; half of them miss the thinned index), 60 seeded chimeras of
         unitig pieces of the dump (several colour sets per read) + 4 edge cases
Outputs: s4546small_reads.fa, s4546small_full_intersection.tsv.gz, s4546small_threshold_union_{0.8,0.3}.tsv.gz in the
         reference's ascii output format "<id>\\t<count>[\\t<colour>...]" (src/ps_utils.cpp:55-71), gzip-compressed
Oracle : oracle/dump_oracle.py — colour set of a k-mer = colour set of its unitig, straight from the dump text; shares no code
         with the engine or with oracle/fulgor_oracle.hpp (the restatement is CHECKED against these vectors, as the engine is)."""
import gzip
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import fulgor_amd
    from conftest import DATA, S10_GENOMES
    from fulgor_amd import synth
    from fulgor_amd.reads import ReadGenerator
    from oracle.dump_oracle import DumpOracle
    fg, extra = synth.ensure_s4546_small(DATA, S10_GENOMES)
    base = os.path.join(DATA, "s4546small_dump")
    if not os.path.exists(base + ".unitigs.fa"):
        ix = fulgor_amd.Index(fg, device=-1)
        ix.dump(base)
        ix.close()
    gen = ReadGenerator(S10_GENOMES[:1], raw_sequences=extra)
    b, o = gen.generate(0, 96, 150, 4546)
    reads = [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(96)]
    # 60 chimeras of two or three unitig pieces of the dump itself (every read meets several colour sets), seeded
    import numpy as np
    rng = np.random.default_rng(4546)
    unitigs = [l.strip() for l in open(base + ".unitigs.fa", "rb") if not l.startswith(b">")]
    longer = [u for u in unitigs if len(u) >= 70]
    for _ in range(60):
        parts = []
        for _ in range(int(rng.integers(2, 4))):
            u = longer[int(rng.integers(len(longer)))]
            st = int(rng.integers(0, len(u) - 60))
            parts.append(u[st:st + int(rng.integers(45, 75))])
        r = bytearray(b"".join(parts)[:150])
        if rng.integers(3) == 0:
            r[int(rng.integers(len(r)))] = b"ACGT"[int(rng.integers(4))]
        reads.append(bytes(r))
    reads += [reads[3][:30], reads[5][:75] + b"N" + reads[5][76:], reads[7].lower(), reads[11] + reads[12]]
    with open(os.path.join(HERE, "s4546small_reads.fa"), "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">r%d\n%s\n" % (i, r))
    orc = DumpOracle(base)

    def dump(name, fn):
        with gzip.GzipFile(os.path.join(HERE, name), "wb", mtime=0) as f:
            for i, r in enumerate(reads):
                cols = fn(r)
                f.write(("\t".join([str(i), str(len(cols))] + [str(c) for c in cols]) + "\n").encode())

    dump("s4546small_full_intersection.tsv.gz", orc.full_intersection)
    for tau in (0.8, 0.3):
        dump("s4546small_threshold_union_%s.tsv.gz" % tau, lambda r, t=tau: orc.threshold_union(r, t))
    print("distinct canonical 31-mers:", len(orc.keys), "colour sets:", len(orc.sets))


if __name__ == "__main__":
    main()
