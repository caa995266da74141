"""ORACLE — test infrastructure only. First-principles oracle over the reference's DUMP files (`fulgor dump`,
src/index.cpp:59-120): <base>.unitigs.fa ("> color_set_id=<id>" headers), <base>.color_sets.txt ("size=<n> c0 c1 ..."),
<base>.metadata.txt. No unitig walking, no minimizers, no compressed lists, no iterators; shares no code with the engine or
with oracle/fulgor_oracle.hpp (only the canonical k-mer routine of oracle/kmer_oracle.py):

  colour set of a k-mer    = the colour set of the unitig that contains it (either strand)          
  full-intersection(read)  = intersection of the colour sets of the read's positive k-mers (empty if none)
                             (src/ps_full_intersection.cpp:334-400, App. B.4)
  threshold-union(read,t)  = colours c with #positive k-mers whose set contains c >= uint64(double(P) * t)
                             (src/ps_threshold_union.cpp:320-402, include/util.hpp:160-208)

Used by tests/golden/make_golden_s4546small.py to produce golden vectors at 4546 colours."""
import numpy as np

from oracle.kmer_oracle import canonical_kmers


class DumpOracle:
    def __init__(self, base):
        meta = dict(line.strip().split("=") for line in open(base + ".metadata.txt") if "=" in line)
        self.k = int(meta["k"])
        self.n = int(meta["num_colors"])
        self.sets = []
        with open(base + ".color_sets.txt") as f:
            for line in f:
                t = line.split()
                cols = np.array(t[1:], dtype=np.int64)
                assert t[0] == "size=%d" % len(cols) and (np.diff(cols) > 0).all()
                self.sets.append(cols)
        keys, ids = [], []
        sid = None
        with open(base + ".unitigs.fa", "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    sid = int(line.split(b"color_set_id=")[1])
                else:
                    km, ok = canonical_kmers(line.strip(), self.k)
                    assert ok.all()
                    keys.append(km)
                    ids.append(np.full(len(km), sid, dtype=np.int64))
        keys, ids = np.concatenate(keys), np.concatenate(ids)
        order = np.argsort(keys, kind="stable")
        self.keys, self.ids = keys[order], ids[order]
        assert (self.keys[1:] != self.keys[:-1]).all(), "a k-mer occurs in two unitigs"

    def positive_set_ids(self, read):
        km, ok = canonical_kmers(read, self.k)
        if len(km) == 0:
            return np.zeros(0, dtype=np.int64)
        idx = np.minimum(np.searchsorted(self.keys, km), len(self.keys) - 1)
        hit = ok & (self.keys[idx] == km)
        return self.ids[idx[hit]]

    def full_intersection(self, read):
        ids = self.positive_set_ids(read)
        if len(ids) == 0:
            return []
        out = None
        for s in np.unique(ids):
            out = self.sets[s] if out is None else np.intersect1d(out, self.sets[s], assume_unique=True)
        return out.tolist()

    def threshold_union(self, read, tau):
        ids = self.positive_set_ids(read)
        if len(ids) == 0:
            return []
        score = np.zeros(self.n, dtype=np.int64)
        u, m = np.unique(ids, return_counts=True)
        for s, c in zip(u, m):
            score[self.sets[s]] += c
        return np.nonzero(score >= int(float(len(ids)) * tau))[0].tolist()
