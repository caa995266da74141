"""A seeded synthetic collection of 256 small genomes (test data; golden vectors above 64 colours).

An ancestor of 24 kb of random sequence evolves down a random binary phylogeny over 256 leaves: every edge applies
point substitutions (0.15 % of the sites) and, now and then, gains a 400-base accessory segment or loses a stretch of
300 bases. Colour id = leaf number. Everything derives from numpy's PCG64 with a fixed seed, so tests and the
generator of the golden files (make_golden_c256.py) build the very same genomes."""
import numpy as np

N_GENOMES = 256
SEED = 256031


def genomes(seed=SEED, n=N_GENOMES, length=24000):
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    root = alpha[rng.integers(0, 4, length)]
    out = []

    def evolve(seq):
        seq = seq.copy()
        nsub = max(1, int(len(seq) * 0.0015))
        pos = rng.integers(0, len(seq), nsub)
        seq[pos] = alpha[rng.integers(0, 4, nsub)]
        u = rng.random()
        if u < 0.25:  # gain
            p = int(rng.integers(0, len(seq)))
            seq = np.concatenate((seq[:p], alpha[rng.integers(0, 4, 400)], seq[p:]))
        elif u < 0.40 and len(seq) > 5000:  # loss
            p = int(rng.integers(0, len(seq) - 300))
            seq = np.concatenate((seq[:p], seq[p + 300:]))
        return seq

    def split(seq, leaves):
        if leaves == 1:
            out.append(seq.tobytes())
            return
        left = int(min(leaves - 1, max(1, round(leaves * (0.3 + 0.4 * rng.random())))))
        split(evolve(seq), left)
        split(evolve(seq), leaves - left)

    split(root, n)
    assert len(out) == n
    return out


def reads(gen, count=400, length=150, seed=SEED + 1):
    """seeded reads: 90 % drawn from a genome (either strand, 1 % substitutions), 10 % random; plus edge cases"""
    rng = np.random.default_rng(seed)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    out = []
    for _ in range(count):
        if rng.random() < 0.1:
            out.append(bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, length)]))
            continue
        g = gen[int(rng.integers(0, len(gen)))]
        p = int(rng.integers(0, len(g) - length))
        r = bytearray(g[p:p + length])
        for q in np.flatnonzero(rng.random(length) < 0.01):
            r[q] = b"ACGT"[int(rng.integers(0, 4))]
        r = bytes(r)
        if rng.random() < 0.5:
            r = r.translate(comp)[::-1]
        out.append(r)
    g0 = gen[17]
    out += [g0[:30], g0[:31], g0[100:170] + b"N" + g0[171:250], b"", g0[300:450].lower(), g0[1000:1400],
            b"N" * 40 + g0[500:531] + b"N" * 40, b"A" * 150]
    return out


def write_fasta(gen, directory):
    import os
    os.makedirs(directory, exist_ok=True)
    paths = []
    for i, g in enumerate(gen):
        p = os.path.join(directory, "g%03d.fa" % i)
        with open(p, "wb") as f:
            f.write(b">g%03d\n%s\n" % (i, g))
        paths.append(p)
    return paths
