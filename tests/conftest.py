import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data")
S10_GENOMES = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def built():
    from fulgor_amd import _build
    _build.build_all()
    return _build


@pytest.fixture(scope="session")
def s10_dump(built):
    """salmonella_10 in the reference's dump format (built once from the genomes, colour = sorted file order)"""
    base = os.path.join(DATA, "s10")
    if not os.path.exists(base + ".unitigs.fa"):
        os.makedirs(DATA, exist_ok=True)
        subprocess.run([built.BIN_CCDBG, "31", base] + S10_GENOMES, check=True)
    return base


@pytest.fixture(scope="session")
def s10_fgidx(built, s10_dump):
    import fulgor_amd
    p = os.path.join(DATA, "s10.v9.fgidx")
    if not os.path.exists(p):
        ix = fulgor_amd.Index(s10_dump, device=-1)
        ix.save(p)
        ix.close()
    return p


@pytest.fixture(scope="session")
def s10_oracle(built, s10_dump):
    from oracle.pyoracle import OracleIndex
    return OracleIndex.from_dump(s10_dump)


@pytest.fixture(scope="session")
def s10_gpu(built, s10_fgidx):
    import fulgor_amd
    return fulgor_amd.Index(s10_fgidx, device=0)


@pytest.fixture(scope="session")
def c256_dump(built):
    """the seeded 256-genome collection of tests/golden/synth_c256.py in the reference's dump format (colour = genome number)"""
    sys.path.insert(0, GOLDEN)
    import synth_c256
    base = os.path.join(DATA, "c256")
    if not os.path.exists(base + ".unitigs.fa"):
        paths = synth_c256.write_fasta(synth_c256.genomes(), os.path.join(DATA, "c256_genomes"))
        subprocess.run([built.BIN_CCDBG, "31", base] + paths, check=True)
    return base


@pytest.fixture(scope="session")
def s4546small_dump(built):
    """the small synthetic 4546-colour index (seeded, rebuilt identically everywhere) as the reference's dump files, written by
    fgpu_dump on a host-only handle; returns (path of the .fgidx, basename of the dump)"""
    import fulgor_amd
    from fulgor_amd import synth
    fg, _ = synth.ensure_s4546_small(DATA, S10_GENOMES)
    base = os.path.join(DATA, "s4546small_dump")
    if not os.path.exists(base + ".unitigs.fa"):
        ix = fulgor_amd.Index(fg, device=-1)
        ix.dump(base)
        ix.close()
    return fg, base


@pytest.fixture(scope="session")
def c256_oracle(built, c256_dump):
    from oracle.pyoracle import OracleIndex
    return OracleIndex.from_dump(c256_dump)


def load_golden_reads(name="s10_reads.fa"):
    from fulgor_amd.reads import parse_fastx
    return parse_fastx(os.path.join(GOLDEN, name))


def load_golden_tsv(name):
    import gzip
    out = []
    with (gzip.open if name.endswith(".gz") else open)(os.path.join(GOLDEN, name), "rt") as f:
        for line in f:
            t = line.rstrip("\n").split("\t")
            assert int(t[0]) == len(out)
            cols = [int(x) for x in t[2:]]
            assert int(t[1]) == len(cols)
            out.append(cols)
    return out


def csr_to_lists(offs, vals):
    offs = np.asarray(offs, dtype=np.int64)
    return [vals[offs[i]:offs[i + 1]].tolist() for i in range(len(offs) - 1)]


def load_golden_kmer_level():
    """read id -> (positive flags, counts per colour, [(start, num_kmers, colour mask)])"""
    out = {}
    with open(os.path.join(GOLDEN, "s10_kmer_level.tsv")) as f:
        for line in f:
            t = line.rstrip("\n").split("\t")
            flags = np.array([int(c) for c in t[1]], dtype=np.uint8)
            counts = np.array([int(x) for x in t[2].split(",")], dtype=np.uint32) if t[2] else np.zeros(10, dtype=np.uint32)
            runs = [tuple(int(x) for x in r.split(":")) for r in t[3].split(",")] if t[3] else []
            out[int(t[0])] = (flags, counts, runs)
    return out
