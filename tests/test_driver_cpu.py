"""CPU suite part 3: host driver logic — sharding, the hit vector and its all-reduce over 2 gloo ranks
(the N>1 path of bench.py with the oracle standing in for the kernels), formatters, CLI argument errors."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden_reads
from fulgor_amd import driver, pack_reads


def test_shard_ranges_are_contiguous_and_cover():
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            rs = [driver.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            assert all(a <= b for a, b in rs)


def test_ascii_formatter_matches_reference_formatter(s10_oracle):
    reads = load_golden_reads()[:200]
    b, o = pack_reads(reads)
    offs, cols = s10_oracle.full_intersection(b, o)
    assert driver.format_ascii(17, offs, cols) == s10_oracle.format_ascii(offs, cols, first_id=17)
    # and the golden file itself is in that format
    want = open(os.path.join(GOLDEN, "s10_full_intersection.tsv"), "rb").read().splitlines(keepends=True)[:200]
    assert driver.format_ascii(0, offs, cols) == b"".join(want)


def test_compressed_formatter_matches_oracle_and_roundtrips(s10_oracle):
    """byte-identical with the restated psa_compressed_formatter, across block boundaries and batches,
    for n = 10 and for a 4546-colour universe (all three record encodings)"""
    from oracle import pyoracle
    reads = load_golden_reads()
    b, o = pack_reads(reads)
    offs, cols = s10_oracle.full_intersection(b, o)
    f = driver.Formatter("compressed", 10)
    half = 400
    o1 = offs[:half + 1]
    o2 = offs[half:] - offs[half]
    got = f.header + f.add(0, o1, cols[:int(offs[half])]) + f.add(half, o2, cols[int(offs[half]):]) + f.finish()
    assert got == pyoracle.format_compressed(offs, cols, 10)
    ids, po, pc = pyoracle.parse_compressed(got)
    assert ids.tolist() == list(range(len(reads))) and np.array_equal(po, offs) and np.array_equal(pc, cols)
    rng = np.random.default_rng(1)
    sizes = np.concatenate([rng.integers(0, 40, 300), rng.integers(1200, 3300, 40), rng.integers(3500, 4547, 40), [0, 4546]])
    rng.shuffle(sizes)
    lists = [np.sort(rng.choice(4546, size=s, replace=False)).astype(np.uint32) for s in sizes]
    offs = np.zeros(len(lists) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(sizes)
    cols = np.concatenate(lists)
    f = driver.Formatter("compressed", 4546)
    got = f.header + f.add(7, offs, cols) + f.finish()
    assert got == pyoracle.format_compressed(offs, cols, 4546, first_id=7)
    ids, po, pc = pyoracle.parse_compressed(got)
    assert ids.tolist() == list(range(7, 7 + len(lists))) and np.array_equal(po, offs) and np.array_equal(pc, cols)
    assert got.count(b"") and len(got) > (1 << 14)  # several blocks


def test_binary_formatter_layout():
    offs = np.array([0, 2, 2], dtype=np.uint64)
    cols = np.array([5, 9], dtype=np.uint32)
    raw = driver.format_binary(3, offs, cols)
    assert np.frombuffer(raw, dtype="<u4").tolist() == [3, 2, 5, 9, 4, 0]


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from fulgor_amd import driver, pack_reads
from fulgor_amd.reads import parse_fastx
from oracle.pyoracle import OracleIndex
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
reads = parse_fastx(os.path.join(sys.argv[1], "tests", "golden", "s10_reads.fa"))
lo, hi = driver.shard_range(len(reads), rank, world)
orc = OracleIndex.from_dump(os.path.join(sys.argv[1], "data", "s10"))
b, o = pack_reads(reads[lo:hi])
offs, cols = orc.full_intersection(b, o, threads=2)
t = torch.from_numpy(driver.hit_vector(offs, cols, 10))
driver.all_reduce_hits(t)
if rank == 0:
    np.save(sys.argv[2], t.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_hit_reduction(s10_oracle, s10_dump, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "hits.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                    "--master-addr", "127.0.0.1", "--master-port", "29541", str(script), ROOT, str(out)],
                   check=True, env=env, timeout=600)
    reads = load_golden_reads()
    b, o = pack_reads(reads)
    offs, cols = s10_oracle.full_intersection(b, o)
    want = driver.hit_vector(offs, cols, 10)
    assert np.array_equal(np.load(out), want)
    assert want[10] == len(reads)


def test_cli_argument_errors():
    from fulgor_amd import cli
    assert cli.main(["pseudoalign", "-i", "x", "-q", "y", "-o", "z", "-r", "1.5"]) == 1
    assert cli.main(["pseudoalign", "-i", "x", "-q", "y", "-o", "z", "-r", "0.5", "--deduplicate"]) == 1
    assert cli.main(["pseudoalign", "-i", "x", "-q", "y", "-o", "z", "--format", "weird"]) == 1
    assert cli.main(["pseudoalign", "-i", "x"]) == 1
    with pytest.raises(ValueError):
        driver.Formatter("weird", 10)
    assert cli.main(["build"]) == 1


def test_dedup_temp_file_formats_roundtrip(tmp_path):
    """the reference's two --deduplicate temp-file layouts (tools/pseudoalign.cpp:91-226, ps_utils.cpp:327-369)"""
    import struct
    from fulgor_amd import driver
    ido = np.array([0, 2, 2, 5, 7, 10], dtype=np.uint64)  # reads 0..4; read 1 has no ids; reads 0 and 3 share a list
    ids = np.array([4, 9, 1, 2, 3, 4, 9, 1, 2, 8], dtype=np.uint32)
    p1 = str(tmp_path / "fetch.tmp")
    driver.write_fetched_ids(p1, ido[:3], ids[:2], first_read_id=0)     # two appends, like two worker flushes
    driver.write_fetched_ids(p1, ido[2:] - ido[2], ids[2:], first_read_id=2)
    want = struct.pack("<2I2I", 0, 2, 4, 9) + struct.pack("<2I", 1, 0) + struct.pack("<2I3I", 2, 3, 1, 2, 3) + \
        struct.pack("<2I2I", 3, 2, 4, 9) + struct.pack("<2I3I", 4, 3, 1, 2, 8)
    assert open(p1, "rb").read() == want
    rid, o2, i2 = driver.read_fetched_ids(p1)
    assert rid.tolist() == [0, 1, 2, 3, 4] and np.array_equal(o2, ido) and np.array_equal(i2, ids)
    unmapped, recs = driver.deduplicate_fetched(rid, o2, i2)
    assert unmapped == [1]
    assert recs == [(2, (1, 2, 3)), (4, (1, 2, 8)), (0, (4, 9)), (3, None)]
    p2 = str(tmp_path / "dedup.tmp")
    driver.write_preprocessed(p2, recs)
    assert open(p2, "rb").read() == struct.pack("<5I", 4, 2, 1, 2, 3) + struct.pack("<5I", 4, 4, 1, 2, 8) + \
        struct.pack("<4I", 3, 0, 4, 9) + struct.pack("<2I", 1, 3)
    got = [(r, l.tolist()) for b in driver.read_preprocessed(p2, batch=3) for r, l in b]
    assert got == [(2, [1, 2, 3]), (4, [1, 2, 8]), (0, [4, 9]), (3, [4, 9])]


def test_native_fastx_reader_matches_python_parser(built, tmp_path):
    """fgpu_fastx_* (kseq semantics: multi-line FASTA and FASTQ, CRLF, quality lines starting with '@', missing final
    newline, gzip or plain) against the simple Python parser and hand-made expectations; batches keep file order"""
    import gzip
    from fulgor_amd.reads import FastxReader, parse_fastx
    rng = np.random.default_rng(3)
    seqs = ["".join("ACGTN"[c] for c in rng.integers(0, 5, size=int(l))) for l in rng.integers(0, 400, size=3000)]
    fq = "".join("@r%d some text\n%s\n+\n%s\n" % (i, s, "@" * len(s)) for i, s in enumerate(seqs))  # '@' qualities
    fa = "".join(">s%d\r\n%s" % (i, "".join(s[j:j + 60] + "\r\n" for j in range(0, len(s), 60))) for i, s in enumerate(seqs))
    files = {"a.fq": fq.encode(), "b.fa": fa.encode(), "c.fq": fq.encode()[:-1]}  # c: no final newline
    for name, data in files.items():
        for gz in (False, True):
            p = str(tmp_path / (name + (".gz" if gz else "")))
            with (gzip.open if gz else open)(p, "wb") as f:
                f.write(data)
            got = []
            for bases, offs in FastxReader(p, batch=70000):  # reader chunks are 65536 reads: batches of one chunk
                b = bytes(bases)
                got += [b[int(offs[i]):int(offs[i + 1])].decode() for i in range(len(offs) - 1)]
            assert got == seqs, name
    # multi-line FASTQ (kseq accepts it): sequence and quality wrapped at 50
    ml = "".join("@m%d\n%s+\n%s" % (i, "".join(s[j:j + 50] + "\n" for j in range(0, len(s), 50)) or "\n",
                                     "".join("I" * len(s[j:j + 50]) + "\n" for j in range(0, len(s), 50)) or "\n")
                 for i, s in enumerate(seqs[:200]))
    p = str(tmp_path / "ml.fq")
    open(p, "w").write(ml)
    got = []
    for bases, offs in FastxReader(p):
        b = bytes(bases)
        got += [b[int(offs[i]):int(offs[i + 1])].decode() for i in range(len(offs) - 1)]
    assert got == seqs[:200]
    assert [s.decode() for s in parse_fastx(str(tmp_path / "a.fq.gz"))] == seqs  # the Python parser agrees on plain 4-line FASTQ
    rd = FastxReader(str(tmp_path / "a.fq.gz"), batch=70000)  # record names: the header up to the first blank
    names = []
    for _ in rd:
        names += rd.names()
    assert names == ["r%d" % i for i in range(len(seqs))]
    with pytest.raises(RuntimeError):
        FastxReader(str(tmp_path / "missing.fq"))


def test_parallel_reader_ranges_parts_and_batch_limit(built, tmp_path):
    """plain files are parsed range by range by several threads (8 MB ranges: a 40 MB FASTQ has five), parts of a file
    partition its records whatever the cut points, and a batch never holds more than max_reads reads"""
    from fulgor_amd.reads import FastxReader, count_reads
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seqs = [bytes(alpha[rng.integers(0, 5, size=int(l))]) for l in rng.integers(100, 220, size=120000)]
    quals = [bytes(rng.integers(33, 75, size=len(s), dtype=np.uint8)) for s in seqs]  # '@', '+' and '>' occur at line starts ...
    quals = [(b">" if i % 3 == 0 else b"@" if i % 3 == 1 else b"+") + q[1:] for i, q in enumerate(quals)]  # ... of every quality line
    p = str(tmp_path / "big.fq")
    with open(p, "wb") as f:
        for i, (s, q) in enumerate(zip(seqs, quals)):
            f.write(b"@read%d/1\n%s\n+\n%s\n" % (i, s, q))
    size = os.path.getsize(p)
    assert size > 36 << 20

    def collect(**kw):
        got, sizes = [], []
        rd = FastxReader(p, copy=True, **kw)
        for bases, offs in rd:
            b = bytes(bases)
            got += [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
            sizes.append(len(offs) - 1)
        rd.close()
        return got, sizes

    got, sizes = collect(batch=4096, threads=6)
    assert got == seqs and max(sizes) <= 4096 and sizes[:-1] == [4096] * (len(sizes) - 1)
    got, sizes = collect(batch=1 << 20, threads=2)
    assert got == seqs and len(sizes) == 1
    cuts = [0, size // 3 + 17, size // 3 + 18, (2 * size) // 3, size]  # arbitrary byte positions, one range nearly empty
    total, joined = 0, []
    for a, b in zip(cuts[:-1], cuts[1:]):
        part, _ = collect(batch=50000, threads=3, begin=a, end=b)
        assert count_reads(p, a, b, 3) == len(part)
        joined += part
        total += len(part)
    assert total == len(seqs) and joined == seqs
    # cut points inside the sequence line of records whose quality line begins with '>' / '@' / '+': the first line start
    # behind such a cut that LOOKS like a header is that quality line; the part must begin at the record after it
    starts = np.cumsum([0] + [len(b"@read%d/1\n" % i) + 2 * len(s) + 4 for i, s in enumerate(seqs)])
    picks = [int(i) for i in rng.choice(len(seqs) - 1, size=30, replace=False)]
    cuts = sorted(int(starts[i]) + len(b"@read%d/1\n" % i) + 10 for i in picks)
    cuts = [0] + cuts + [size]
    joined = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        joined += collect(batch=50000, threads=2, begin=a, end=b)[0]
    assert joined == seqs
    # every quality line begins with '>' (Phred 29): in a FASTQ file that is not a FASTA header, wherever a part is cut
    small = [b"ACGT" * 30] * 2000
    p2 = str(tmp_path / "gt.fq")
    with open(p2, "wb") as f:
        for i, sq in enumerate(small):
            f.write(b"@r%d\n%s\n+\n>%s\n" % (i, sq, b"I" * (len(sq) - 1)))
    sz = os.path.getsize(p2)
    for cut in range(120000, 120000 + 300, 7):  # cut points in every kind of line
        parts = []
        for a, b in ((0, cut), (cut, sz)):
            rd = FastxReader(p2, copy=True, batch=5000, threads=2, begin=a, end=b)
            for bases, offs in rd:
                bb = bytes(bases)
                parts += [bb[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
            rd.close()
        assert parts == small


def _bgzf(data, rng, max_block=65280):
    """block-compressed gzip as bgzip / htslib write it (SAM specification 4.1): members of at most 64 KB with their total size
    in a 'BC' extra subfield, and the empty end-of-file member"""
    import struct, zlib
    out, at = [], 0
    while True:
        n = min(len(data) - at, int(rng.integers(1, max_block + 1)))
        blk = data[at:at + n]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        cd = co.compress(blk) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd +
                   struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk)))
        at += n
        if n == 0:
            break
    return b"".join(out)


def test_ordinary_gzip_whole_and_streamed(built, tmp_path, monkeypatch):
    """an ordinary gzip file of moderate size is inflated in one go (libdeflate, when the host has it) and parsed by the pool;
    otherwise, and for large files, zlib streams it. Same records either way: one member, several members, zero padding
    behind the last member; a truncated file is an error either way"""
    import gzip
    from fulgor_amd.reads import FastxReader
    rng = np.random.default_rng(8)
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seqs = [bytes(alpha[rng.integers(0, 5, size=int(l))]) for l in rng.integers(50, 300, size=70000)]
    text = b"".join(b"@q%d\n%s\n+\n%s\n" % (i, s, b">" + b"F" * (len(s) - 1)) for i, s in enumerate(seqs))
    half = text.index(b"\n@q35000\n") + 1
    files = {"one.fq.gz": gzip.compress(text, 1), "two.fq.gz": gzip.compress(text[:half], 1) + gzip.compress(text[half:], 6),
             "pad.fq.gz": gzip.compress(text, 1) + b"\0" * 700}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)

    def collect(path):
        got = []
        rd = FastxReader(str(path), copy=True, batch=30000, threads=4)
        for bases, offs in rd:
            b = bytes(bases)
            got += [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
        rd.close()
        return got

    (tmp_path / "cut.fq.gz").write_bytes(files["one.fq.gz"][:len(files["one.fq.gz"]) // 2])
    for stream in (False, True):
        if stream:
            monkeypatch.setenv("FULGOR_GZIP_STREAM", "1")
        for name in ("one.fq.gz", "two.fq.gz") + (() if stream else ("pad.fq.gz",)):  # (zlib's gzread stops at the padding as well)
            assert collect(tmp_path / name) == seqs, (name, stream)
        with pytest.raises(RuntimeError):
            collect(tmp_path / "cut.fq.gz")


def test_block_compressed_gzip_is_read_in_parallel(built, tmp_path):
    """BGZF (bgzip) files are gzip files whose members can be inflated independently: the reader does so on all threads and
    parses range by range as for a plain file. Same records as the plain file, whatever the member sizes (records, lines and
    range boundaries fall anywhere inside members); an ordinary gzip reader sees the same stream; damage is reported."""
    import gzip
    from fulgor_amd.reads import FastxReader
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    seqs = [bytes(alpha[rng.integers(0, 5, size=int(l))]) for l in rng.integers(100, 220, size=90000)]
    quals = [bytes(rng.integers(33, 75, size=len(s), dtype=np.uint8)) for s in seqs]
    quals = [(b">" if i % 3 == 0 else b"@" if i % 3 == 1 else b"+") + q[1:] for i, q in enumerate(quals)]  # line starts that look like headers
    plain = b"".join(b"@read%d extra\n%s\n+\n%s\n" % (i, s, q) for i, (s, q) in enumerate(zip(seqs, quals)))
    assert len(plain) > 26 << 20  # four ranges of 8 MB

    def collect(path, **kw):
        got, names = [], []
        rd = FastxReader(path, copy=True, **kw)
        for bases, offs in rd:
            b = bytes(bases)
            got += [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
            names += rd.names()
        rd.close()
        return got, names

    for tag, mb in (("a", 65280), ("b", 3000)):  # full-size members, and many small ones of random sizes
        p = str(tmp_path / ("reads_%s.fq.gz" % tag))
        comp = _bgzf(plain, rng, mb)
        with open(p, "wb") as f:
            f.write(comp)
        assert gzip.decompress(comp) == plain  # (what the reference's gzip reader would see)
        got, names = collect(p, batch=20000, threads=5)
        assert got == seqs and names == ["read%d" % i for i in range(len(seqs))]
        got, _ = collect(p, batch=1 << 20, threads=1)
        assert got == seqs
    # parts of the inflated text, as the ranks of a multi-GPU run take them (positions in the text, not in the file)
    from fulgor_amd.reads import count_reads, text_size
    assert text_size(p) == (len(plain), True)
    cuts = [0, len(plain) // 3 + 5, len(plain) // 3 + 6, (2 * len(plain)) // 3, len(plain)]
    joined = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        part, _ = collect(p, batch=50000, threads=3, begin=a, end=b)
        assert count_reads(p, a, b, 3) == len(part)
        joined += part
    assert joined == seqs
    pz = str(tmp_path / "stream.fq.gz")
    with gzip.open(pz, "wb", compresslevel=1) as f:
        f.write(plain[:1 << 20])
    assert text_size(pz) == (0, False) and text_size(str(tmp_path / "reads_a.fq.gz"))[1]
    with pytest.raises(RuntimeError):
        FastxReader(pz, begin=100, end=5000)
    fa = b"".join(b">s%d\n%s\n%s\n" % (i, s[:60], s[60:]) for i, s in enumerate(seqs[:5000]))  # wrapped FASTA
    p = str(tmp_path / "seqs.fa.gz")
    with open(p, "wb") as f:
        f.write(_bgzf(fa, rng, 20000))
    got, _ = collect(p, batch=1000, threads=3)
    assert got == seqs[:5000]
    # records far longer than the megabyte the boundary search inflates first (long reads): the window must grow
    longs = [bytes(alpha[rng.integers(0, 4, size=int(l))]) for l in (2_600_000, 50, 1_300_000, 3_100_000, 700, 2_200_000)]
    for kind in ("fq", "fa"):
        if kind == "fq":
            text = b"".join(b"@L%d\n%s\n+\n%s\n" % (i, s, b"@" + b"I" * (len(s) - 1)) for i, s in enumerate(longs))
        else:
            text = b"".join(b">L%d\n%s\n" % (i, s) for i, s in enumerate(longs))
        p = str(tmp_path / ("long.%s.gz" % kind))
        with open(p, "wb") as f:
            f.write(_bgzf(text, rng, 65280))
        assert collect(p, batch=4, threads=4)[0] == longs
        cuts = sorted(set([0, len(text)] + [int(x) for x in rng.integers(0, len(text), size=6)]))
        joined = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            joined += collect(p, batch=4, threads=2, begin=a, end=b)[0]
        assert joined == longs, (kind, cuts)
    with open(str(tmp_path / "empty.fq.gz"), "wb") as f:
        f.write(_bgzf(b"", rng))
    assert collect(str(tmp_path / "empty.fq.gz"), batch=10)[0] == []
    bad = bytearray(comp)
    bad[len(bad) // 2] ^= 0x55  # a flipped byte in the middle of the file
    with open(str(tmp_path / "bad.fq.gz"), "wb") as f:
        f.write(bytes(bad))
    with pytest.raises(RuntimeError):
        collect(str(tmp_path / "bad.fq.gz"), batch=20000, threads=4)


MULTI_RANK_CLI_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch.distributed as dist
from fulgor_amd import driver
from oracle.pyoracle import OracleIndex

class Reads:
    def __init__(self, b, o): self.b, self.o = np.array(b), np.array(o)
    def close(self): pass

class Result:
    """stands in for the GPU result of a pass in this CPU test of the sharding logic: computed by the oracle"""
    def __init__(self, orc): self.orc = orc
    def sizes(self): return len(self.o) - 1, len(self.c), int((np.diff(self.o.astype(np.int64)) > 0).sum())
    def format_view(self, fmt, first_id): return driver.format_ascii(first_id, self.o, self.c)
    def close(self): pass

class Index:
    def __init__(self, base): self.orc = OracleIndex.from_dump(base)
    def num_colors(self): return 10
    def new_result(self): return Result(self.orc)
    def upload_reads(self, b, o): return Reads(b, o)
    def run(self, reads, res, algo, tau): res.o, res.c = self.orc.full_intersection(reads.b, reads.o, threads=2)

# every rank opens its part of the query file ONCE (one parse; the record count comes from a boundary walk on the same handle)
from fulgor_amd import _native
L = _native.lib()
opens, real_open, real_count = [], L.fgpu_fastx_open_part, L.fgpu_fastx_count
def counting_open(*a):
    opens.append(a[0])
    return real_open(*a)
def no_second_parse(*a):
    raise AssertionError("fgpu_fastx_count parses the part a second time")
L.fgpu_fastx_open_part, L.fgpu_fastx_count = counting_open, no_second_parse

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, mapped = driver.pseudoalign_sharded(lambda: Index(os.path.join(sys.argv[1], "data", "s10")), sys.argv[2], sys.argv[3],
                                       rank=rank, world=world, io_threads=2, batch=300)
assert len(opens) == 1, opens
if rank == 0:
    open(sys.argv[3] + ".counters", "w").write("%d %d" % (n, mapped))
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_cli_path_on_gloo(s10_dump, tmp_path, world):
    """driver.pseudoalign_sharded, the body of `pseudoalign --gpus N`, launched by torchrun on gloo: every rank takes a byte range
    of the query file, numbers its reads from the all-gathered counts, writes its part; the parts are joined in rank order
    and the counters all-reduced. The passes themselves are computed by the oracle here (no GPU in this suite); the
    result must be the golden ascii output, byte for byte."""
    script = tmp_path / "worker.py"
    script.write_text(MULTI_RANK_CLI_WORKER)
    reads = load_golden_reads()
    q = tmp_path / "reads.fq"
    with open(q, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
    out = tmp_path / "out.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                    "--master-addr", "127.0.0.1", "--master-port", str(29550 + world), str(script), ROOT, str(q), str(out)],
                   check=True, env=env, timeout=600)
    want = open(os.path.join(ROOT, "tests", "golden", "s10_full_intersection.tsv"), "rb").read()
    assert open(out, "rb").read() == want
    n, mapped = map(int, open(str(out) + ".counters").read().split())
    assert n == len(reads) and mapped == sum(1 for l in want.splitlines() if l.split(b"\t")[1] != b"0")
    assert not [p for p in os.listdir(tmp_path) if ".part" in p]
    if world == 2:  # the same query file compressed in blocks (bgzip): the ranks take parts of the inflated text
        qz = tmp_path / "reads.fq.gz"
        qz.write_bytes(_bgzf(q.read_bytes(), np.random.default_rng(1), 4000))
        out2 = tmp_path / "out_bgzf.txt"
        subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29561", str(script), ROOT, str(qz), str(out2)],
                       check=True, env=env, timeout=600)
        assert open(out2, "rb").read() == want


@pytest.mark.parametrize("piece_kb", [None, "1"])
def test_reader_parts_fuzz(built, tmp_path, monkeypatch, piece_kb):
    """(piece_kb: plain FASTQ records of a range are read and parsed in pieces of half a megabyte; pieces of 1 KB put a piece end
    into nearly every record of these small files)
    seeded fuzz of the range logic: 4-line FASTQ and single- or multi-line FASTA with every nasty line start (quality lines
    beginning with '@', '>' or '+', names with blanks, CRLF, no final newline, empty sequences in FASTA), cut into parts at
    random positions: the parts must partition the records, names included, whatever the cuts"""
    from fulgor_amd.reads import FastxReader, count_reads
    if piece_kb:
        monkeypatch.setenv("FULGOR_READER_PIECE_KB", piece_kb)
    rng = np.random.default_rng(2024)
    alpha = np.frombuffer(b"ACGTNacgt", dtype=np.uint8)
    qual_first = b"@>+!I5#"

    def collect(path, **kw):
        seqs, names = [], []
        rd = FastxReader(path, copy=True, batch=37, threads=3, **kw)
        counted = rd.count()  # (fgpu_fastx_count_part: the grammar walked without copies, on the same handle, before anything is consumed)
        for bases, offs in rd:
            b = bytes(bases)
            seqs += [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
            names += rd.names()
        rd.close()
        assert counted == len(seqs), (path, kw, counted, len(seqs))
        return seqs, names

    for trial in range(600 if piece_kb else 1500):
        fastq = trial % 2 == 0 or bool(piece_kb)
        eol = b"\r\n" if trial % 5 == 3 else b"\n"
        n = int(rng.integers(1, 120))
        want_s, want_n, text = [], [], []
        for i in range(n):
            ln = int(rng.integers(1 if fastq else 0, 90))
            s = bytes(alpha[rng.integers(0, len(alpha), size=ln)])
            name = b"rec%d_%d" % (trial, i)
            hdr = name + (b" some comment" if i % 3 == 0 else b"")
            if fastq:
                q = bytes([qual_first[int(rng.integers(0, len(qual_first)))]]) + bytes(rng.integers(33, 75, size=ln - 1, dtype=np.uint8))
                text.append(b"@" + hdr + eol + s + eol + (b"+" + (name if i % 4 == 0 else b"")) + eol + q + eol)
            else:
                w = int(rng.integers(1, 100))
                lines = [s[j:j + w] for j in range(0, len(s), w)] if trial % 4 == 1 else [s]
                text.append(b">" + hdr + eol + b"".join(l + eol for l in lines))
            want_s.append(s)
            want_n.append(name.decode())
        data = b"".join(text)
        if trial % 7 == 0 and data.endswith(eol):
            data = data[:-len(eol)]  # no newline at the end of the file
        p = str(tmp_path / ("f%d.%s" % (trial, "fq" if fastq else "fa")))
        with open(p, "wb") as f:
            f.write(data)
        assert collect(p) == (want_s, want_n), trial
        cuts = sorted(set([0, len(data)] + [int(x) for x in rng.integers(0, len(data) + 1, size=int(rng.integers(1, 9)))]))
        js, jn, total = [], [], 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            s_, n_ = collect(p, begin=a, end=b)
            assert count_reads(p, a, b, 2) == len(s_)
            js += s_
            jn += n_
        assert (js, jn) == (want_s, want_n), (trial, cuts)


def test_reader_under_sanitizers(tmp_path):
    """race / memory check of the multi-threaded query reader (SURVEY: race detection): the reader alone, built with
    ThreadSanitizer and with AddressSanitizer + UBSan, reads a plain and a block-compressed FASTQ file (whole and in parts)
    with a pool of threads; no report, and the same records every time"""
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(ROOT, "tests", "reader_sanitize.cpp")
    inc = os.path.join(ROOT, "fulgor_amd", "csrc")
    rng = np.random.default_rng(1)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    recs = []
    for i in range(60000):
        l = int(rng.integers(80, 200))
        q = bytes([b"@>+I"[i % 4]]) + bytes(rng.integers(35, 74, size=l - 1, dtype=np.uint8))
        recs.append(b"@r%d x\n%s\n+\n%s\n" % (i, bytes(alpha[rng.integers(0, 4, size=l)]), q))
    plain = b"".join(recs)
    (tmp_path / "p.fq").write_bytes(plain)
    (tmp_path / "p.fq.gz").write_bytes(_bgzf(plain, rng, 65280))
    import gzip
    (tmp_path / "s.fq.gz").write_bytes(gzip.compress(plain, 1))
    for tag, flags in (("tsan", ["-fsanitize=thread"]), ("asan", ["-fsanitize=address,undefined"])):
        exe = str(tmp_path / ("h_" + tag))
        r = subprocess.run(["g++", "-O1", "-g", "-std=c++17"] + flags + ["-I", inc, src, "-o", exe, "-lz", "-ldl", "-pthread"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("sanitizer build not available: " + r.stderr[-200:])
        outs = set()
        for f in ("p.fq", "p.fq.gz", "s.fq.gz"):
            for args in (["6"], ["3", "5000000", "14000000"])[:1 if f == "s.fq.gz" else 2]:  # (an ordinary gzip stream has no parts)
                r = subprocess.run([exe, str(tmp_path / f)] + args, capture_output=True, text=True, timeout=600)
                assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
                lines = r.stdout.strip().splitlines()
                assert len(lines) == 3 and len(set(lines)) == 1
                outs.add((tuple(args[1:]), lines[0]))
        # the plain file again in ranges of 256 KB read in pieces of 3 KB (a piece end inside nearly every tenth record)
        r = subprocess.run([exe, str(tmp_path / "p.fq"), "6"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, FULGOR_READER_RANGE_KB="256", FULGOR_READER_PIECE_KB="3"))
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
        outs.add(((), r.stdout.strip().splitlines()[0]))
        assert len(outs) == 2  # plain and block-compressed agree, whole and in parts


def test_wrapped_fastq_is_read_as_one_stream_and_leading_junk_does_not_change_the_kind(built, tmp_path):
    """(1) FASTQ whose sequence and quality lines are wrapped offers no boundary a byte range could start at (a
    quality line that begins with '@', two lines in front of one that begins with '+', looks exactly like a record): such a
    file is recognised at open and parsed as one stream with kseq's grammar — 50000 reads of 250 bases wrapped at 60 columns
    with uniform qualities come out as 50000 reads — and it refuses to be read in parts. (2) A FASTQ file that starts with a
    blank line or stray text is still a FASTQ file: a quality line that begins with '>' is not a FASTA header in any part."""
    from fulgor_amd.reads import FastxReader, count_reads, text_size
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = 50000
    seqs = [bytes(alpha[rng.integers(0, 4, size=250)]) for _ in range(n)]
    p = str(tmp_path / "wrapped.fq")
    with open(p, "wb") as f:
        for i, s in enumerate(seqs):
            q = bytes(rng.integers(33, 74, size=250, dtype=np.uint8))
            f.write(b"@w%d\n" % i + b"".join(s[j:j + 60] + b"\n" for j in range(0, 250, 60)) + b"+\n" +
                    b"".join(q[j:j + 60] + b"\n" for j in range(0, 250, 60)))
    assert os.path.getsize(p) > 24 << 20  # several 8 MB ranges, had it been cut
    got = []
    rd = FastxReader(p, copy=True, batch=20000, threads=4)
    for bases, offs in rd:
        b = bytes(bases)
        got += [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    rd.close()
    assert len(got) == n and got == seqs
    assert text_size(p) == (os.path.getsize(p), False)
    with pytest.raises(RuntimeError):
        count_reads(p, 1000, os.path.getsize(p), 2)
    # the same text block-compressed (bgzip layout): read whole it falls back to the one-stream reader; parts stay an error
    import struct, zlib
    raw = open(p, "rb").read()[:6 << 20]
    raw = raw[:raw.rfind(b"\n@w") + 1]
    nb = raw.count(b"\n@w") + 1
    pz = str(tmp_path / "wrapped.fq.gz")
    with open(pz, "wb") as f:
        for at in list(range(0, len(raw), 60000)) + [len(raw)]:
            blk = raw[at:at + 60000]
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            cd = co.compress(blk) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd +
                    struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk)))
    gotz = []
    rd = FastxReader(pz, copy=True, batch=20000, threads=4)
    for bases, offs in rd:
        b = bytes(bases)
        gotz += [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    rd.close()
    assert len(gotz) == nb and gotz == seqs[:nb]
    with pytest.raises(RuntimeError):
        count_reads(pz, 1000, 3 << 20, 2)
    # (2) leading junk, then strict four-line records whose quality lines all begin with '>'
    small = [b"ACGT" * 30] * 3000
    for junk in (b"\n", b"\xef\xbb\xbf\n", b"# produced by some tool\n\n"):
        p2 = str(tmp_path / "junk.fq")
        with open(p2, "wb") as f:
            f.write(junk)
            for i, sq in enumerate(small):
                f.write(b"@r%d\n%s\n+\n>%s\n" % (i, sq, b"I" * (len(sq) - 1)))
        sz = os.path.getsize(p2)
        assert text_size(p2) == (sz, True)
        for cut in range(200000, 200000 + 260, 13):  # cut points in every kind of line
            parts = []
            for a, b in ((0, cut), (cut, sz)):
                rd = FastxReader(p2, copy=True, batch=5000, threads=2, begin=a, end=b)
                for bases, offs in rd:
                    bb = bytes(bases)
                    parts += [bb[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
                rd.close()
            assert parts == small, (junk, cut)


def test_bench_real_dump_hook_ingests_a_dump_and_labels_the_line(built, tmp_path, monkeypatch):
    """FULGOR_S4546_DUMP (bench.py): the four text files of a `fulgor dump` (src/index.cpp:59-120) take the synthetic index's
    place. Here the dump is the 4546-colour test index written by fgpu_dump: all four files are required, the index is ingested
    into the data directory given, reads are drawn from its unitigs, and the description says what the data is."""
    import fulgor_amd
    from conftest import DATA, S10_GENOMES
    from fulgor_amd import synth
    import bench
    fg, _ = synth.ensure_s4546_small(DATA, S10_GENOMES)
    base = os.path.join(DATA, "s4546small_dump")
    if not os.path.exists(base + ".unitigs.fa"):
        ix = fulgor_amd.Index(fg, device=-1)
        ix.dump(base)
        ix.close()
    # a private copy (links) so that one file can go missing
    mine = str(tmp_path / "salmonella_4546")
    for suffix in (".metadata.txt", ".unitigs.fa", ".color_sets.txt", ".filenames.txt"):
        os.symlink(base + suffix, mine + suffix)
    data = str(tmp_path / "data")
    got_fg, gen, desc = bench.real_dump_workload(mine, data)
    assert got_fg == os.path.join(data, "salmonella_4546.v9.fgidx") and os.path.exists(got_fg)
    assert desc.startswith(bench.REAL_DUMP_PREFIX) and "salmonella_4546" in desc and str(tmp_path) not in desc
    ix = fulgor_amd.Index(got_fg, device=-1)
    assert ix.num_colors() == 4546 and ix.k() == 31
    ix.close()
    b, o = gen.generate(0, 500, 150, 42)
    assert len(o) == 501 and int(o[-1]) == 500 * 150 and set(np.unique(b).tolist()) <= set(b"ACGT")
    b2, _ = gen.generate(0, 500, 150, 42)
    assert np.array_equal(b, b2)  # seeded
    # the environment variable routes the default workload there
    monkeypatch.setenv("FULGOR_S4546_DUMP", mine)
    mtime = os.path.getmtime(got_fg)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))  # (prepare_workload puts its caches under ROOT/data)
    fg2, _, desc2 = bench.prepare_workload("s4546syn")
    assert fg2 == got_fg and desc2 == desc and os.path.getmtime(got_fg) == mtime  # ingested once
    os.remove(mine + ".filenames.txt")
    with pytest.raises(SystemExit, match="filenames.txt is missing"):
        bench.real_dump_workload(mine, data)


def test_reader_mapped_and_read_ranges_agree(built, tmp_path):
    """the ranges of a plain file are read with pread into per-thread buffers, plain FASTQ records in pieces of half a megabyte;
    A/B knobs: FULGOR_READER_MMAP=1 maps the file instead, =2 maps range by range, FULGOR_READER_PIECE_KB sets the piece (0: the
    whole range at once). Same records, same counts, whole and in parts, FASTA and FASTQ (the knobs are read when a reader opens:
    subprocess)."""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    recs = [bytes(alpha[rng.integers(0, 5, size=3000 if i % 5000 == 77 else int(rng.integers(1, 400)))]) for i in range(60000)]  # (a few longer than a piece)
    fq, fa = tmp_path / "r.fq", tmp_path / "r.fa"
    fq.write_bytes(b"".join(b"@q%d x\n%s\n+\n%s\n" % (i, s, b"@" * len(s)) for i, s in enumerate(recs)))
    fa.write_bytes(b"".join(b">q%d\n%s\n" % (i, b"\n".join(s[j:j + 70] for j in range(0, len(s), 70))) for i, s in enumerate(recs)))
    code = r'''
import hashlib, os, sys
sys.path.insert(0, %r)
from fulgor_amd.reads import FastxReader
os.environ["FULGOR_READER_RANGE_KB"] = "256"
for path in sys.argv[1:]:
    size = os.path.getsize(path)
    for a, b in ((0, (1 << 64) - 1), (0, size // 3), (size // 3, size - 1000), (size - 1000, size)):
        h, n = hashlib.sha256(), 0
        rd = FastxReader(path, copy=True, batch=5000, threads=4, begin=a, end=b)
        c = rd.count()
        for bases, offs in rd:
            h.update(bytes(bases)); h.update(offs.tobytes()); n += len(offs) - 1
            h.update("|".join(rd.names()).encode())
        rd.close()
        assert c == n
        print(os.path.basename(path), a, b, n, h.hexdigest())
''' % ROOT
    outs = []
    for mode, piece in (("", ""), ("1", ""), ("2", ""), ("", "0"), ("", "1"), ("", "3"), ("", "64")):
        env = dict(os.environ)
        env.pop("FULGOR_READER_MMAP", None)
        env.pop("FULGOR_READER_PIECE_KB", None)
        if mode:
            env["FULGOR_READER_MMAP"] = mode
        if piece:
            env["FULGOR_READER_PIECE_KB"] = piece
        r = subprocess.run([sys.executable, "-c", code, str(fq), str(fa)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert all(o == outs[0] for o in outs) and outs[0].count("\n") == 8
    assert " 60000 " in outs[0].splitlines()[0]


def test_gzip_of_wrapped_fastq_and_counting_a_block_compressed_part(built, tmp_path):
    """ADVICE r5. (1) An ordinary gzip of a FASTQ file with wrapped lines: with libdeflate the file is inflated whole, and the text
    then cannot be cut into ranges — the reader must fall back to the one-stream reader (kseq's grammar, as the reference reads
    it), not fail to open. (2) Counting the records of a block-compressed part (what every rank of a multi-GPU run does before it
    streams) walks the inflated text a strip at a time and gives it back: the count keeps tens of megabytes resident, not the
    part; the reader then delivers every record; and a count asked for after chunks were handed out is refused, not wrong."""
    import gzip
    from fulgor_amd.reads import FastxReader
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [bytes(alpha[rng.integers(0, 4, size=250)]) for _ in range(2000)]
    text = b"".join(b"@w%d\n" % i + b"".join(s[j:j + 60] + b"\n" for j in range(0, 250, 60)) + b"+\n" +
                    b"".join((b"I" * 250)[j:j + 60] + b"\n" for j in range(0, 250, 60)) for i, s in enumerate(seqs))
    pz = tmp_path / "wrapped.fq.gz"
    pz.write_bytes(gzip.compress(text, 1))
    got = []
    rd = FastxReader(str(pz), copy=True, batch=700, threads=3)
    for bases, offs in rd:
        b = bytes(bases)
        got += [b[int(offs[i]):int(offs[i + 1])] for i in range(len(offs) - 1)]
    rd.close()
    assert got == seqs
    # (2) 1.2 M four-line records, 190 MB of text, block-compressed; in a subprocess so that the resident size is this reader's
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from fulgor_amd.reads import FastxReader
def rss_mb():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGESIZE") / 1e6
path, n = sys.argv[1], int(sys.argv[2])
rd = FastxReader(path, copy=False, batch=100000, threads=4)
r0 = rss_mb()
c = rd.count()
r1 = rss_mb()
assert c == n, (c, n)
assert r1 - r0 < 90, "counting left %%d MB resident" %% (r1 - r0)
total = first = 0
for bases, offs in rd:
    total += len(offs) - 1
    if not first:
        first = 1
        try:
            rd.count()
            raise SystemExit("a count behind the reader was not refused")
        except RuntimeError:
            pass
assert total == n, (total, n)
rd.close()
# a part of the text: counted, then read
rd = FastxReader(path, copy=False, batch=100000, threads=4, begin=50_000_000, end=120_000_000)
c = rd.count()
assert c == sum(len(o) - 1 for _, o in rd)
rd.close()
print("ok", c)
''' % ROOT
    n = 1_200_000
    rows = np.empty((n, 10 + 74 + 3 + 74 + 1), dtype=np.uint8)  # "@rrrrrrrr\n" + 74 bases + "\n+\n" + 74 quality characters + "\n"
    rows[:, 0] = ord("@")
    rows[:, 1:9] = ord("r")
    rows[:, 9] = ord("\n")
    rows[:, 10:84] = alpha[rng.integers(0, 4, size=(n, 74))]
    rows[:, 84:87] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rows[:, 87:161] = ord("@")  # (quality lines that begin with '@')
    rows[:, 161] = ord("\n")
    data = rows.tobytes()
    import zlib, struct
    blocks = []
    for at in range(0, len(data), 65280):  # (zlib level 1: the test is about the reader, not the compressor)
        blk = data[at:at + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        cd = co.compress(blk) + co.flush()
        blocks.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd +
                      struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk)))
    blocks.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00")
    pb = tmp_path / "big.fq.gz"
    pb.write_bytes(b"".join(blocks))
    r = subprocess.run([sys.executable, "-c", code, str(pb), str(n)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout + r.stderr)[-2000:]
