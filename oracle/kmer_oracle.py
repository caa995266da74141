"""ORACLE — test infrastructure only. Independent k-mer-level oracle.

Computes pseudoalignment results straight from the genome collection, with no unitigs, no colour-set
ids, no compressed lists and no code shared with either the product or oracle/fulgor_oracle.hpp:

  colour mask of a k-mer   = set of genomes containing it (either strand), windows with non-ACGT dropped
  full-intersection(read)  = AND of the masks of the read's positive k-mers          (empty if none)
                             == intersection of the distinct colour sets of positive k-mers
                             (src/ps_full_intersection.cpp:334-400)
  threshold-union(read,t)  = colours c with #positive k-mers containing c >= uint64(double(P) * t),
                             P = #positive k-mers (src/ps_threshold_union.cpp:320-402, include/util.hpp:160-208)

Used by tests/golden/make_golden*.py to produce the committed golden vectors. Any number of genomes (masks are
rows of 64-bit words).
"""
import gzip

import numpy as np

_CODE = np.full(256, 4, dtype=np.uint8)
for i, ch in enumerate("ACGT"):
    _CODE[ord(ch)] = i
    _CODE[ord(ch.lower())] = i


def read_fasta(path):
    op = gzip.open if str(path).endswith(".gz") else open
    seqs, cur = [], []
    with op(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur:
                    seqs.append(b"".join(cur))
                cur = []
            else:
                cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    return seqs


def canonical_kmers(seq, k):
    """(canonical k-mer as uint64 with the first base most significant, valid flag) for every window"""
    codes = _CODE[np.frombuffer(seq, dtype=np.uint8)]
    n = len(codes) - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=bool)
    bad = np.concatenate(([0], np.cumsum(codes > 3)))
    valid = (bad[k:] - bad[:-k]) == 0
    c = (codes & 3).astype(np.uint64)
    fw = np.zeros(n, dtype=np.uint64)
    rv = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        fw = (fw << np.uint64(2)) | c[j:j + n]
        rv = rv | ((np.uint64(3) - c[j:j + n]) << np.uint64(2 * j))
    return np.minimum(fw, rv), valid


class KmerOracle:
    def __init__(self, genomes, k=31):
        """genomes: FASTA paths, or lists of contigs (bytes) per genome"""
        self.k = k
        self.n = len(genomes)
        self.W = (self.n + 63) // 64
        keys, owner = [], []
        for g, p in enumerate(genomes):
            ks = []
            for contig in (read_fasta(p) if isinstance(p, str) else p):
                km, ok = canonical_kmers(contig, k)
                ks.append(km[ok])
            u = np.unique(np.concatenate(ks)) if ks else np.zeros(0, dtype=np.uint64)
            keys.append(u)
            owner.append(np.full(len(u), g, dtype=np.uint32))
        keys = np.concatenate(keys)
        owner = np.concatenate(owner)
        order = np.argsort(keys, kind="stable")
        keys, owner = keys[order], owner[order]
        first = np.concatenate(([True], keys[1:] != keys[:-1]))
        self.keys = keys[first]
        row = np.cumsum(first) - 1  # index of the distinct k-mer of every (k-mer, genome) pair
        self.masks = np.zeros((len(self.keys), self.W), dtype=np.uint64)
        np.bitwise_or.at(self.masks, (row, owner >> 6), np.uint64(1) << (owner & 63).astype(np.uint64))

    def kmer_masks(self, read):
        """masks of the read's positive k-mers: array [P, W]"""
        km, ok = canonical_kmers(read, self.k)
        if len(km) == 0 or len(self.keys) == 0:
            return np.zeros((0, self.W), dtype=np.uint64)
        idx = np.searchsorted(self.keys, km)
        idx[idx >= len(self.keys)] = 0
        hit = ok & (self.keys[idx] == km)
        return self.masks[idx[hit]]

    def _colours(self, words):
        return [c for c in range(self.n) if (int(words[c >> 6]) >> (c & 63)) & 1]

    def full_intersection(self, read):
        m = self.kmer_masks(read)
        if len(m) == 0:
            return []
        return self._colours(np.bitwise_and.reduce(m, axis=0))

    def threshold_union(self, read, tau):
        m = self.kmer_masks(read)
        if len(m) == 0:
            return []
        P = len(m)
        min_score = int(float(P) * tau)
        out = []
        for c in range(self.n):
            if int(((m[:, c >> 6] >> np.uint64(c & 63)) & np.uint64(1)).sum()) >= min_score:
                out.append(c)
        return out
