"""Reader rates on one FASTQ file in three forms: plain, ordinary gzip (one stream, one inflating thread), block-compressed
gzip (BGZF: members inflated by the pool of threads). usage: python profiles/gz_reader_rates.py [n reads] [threads]"""
import gzip, os, struct, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fulgor_amd.reads import FastxReader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(3)
rec = np.empty((n, 12 + 150 + 3 + 150 + 1), dtype=np.uint8)
ids = np.arange(n, dtype=np.int64)
rec[:, 0], rec[:, 1], rec[:, 11] = ord("@"), ord("r"), ord("\n")
for d in range(9):
    rec[:, 2 + d] = ord("0") + (ids // 10 ** (8 - d)) % 10
rec[:, 12:162] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 150))]
rec[:, 162:165] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 165:-1] = rng.integers(35, 74, size=(n, 150), dtype=np.uint8)
rec[:, -1] = ord("\n")
plain = rec.tobytes()
del rec
d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
paths = {k: os.path.join(d, "gzr_%d.%s" % (os.getpid(), k)) for k in ("fq", "gz", "bgzf.gz")}
try:
    open(paths["fq"], "wb").write(plain)
    with gzip.open(paths["gz"], "wb", compresslevel=1) as f:
        f.write(plain)
    with open(paths["bgzf.gz"], "wb") as f:
        for at in range(0, len(plain) + 1, 65280):
            blk = plain[at:at + 65280]
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            cd = co.compress(blk) + co.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd +
                    struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk)))
        if len(plain) % 65280 != 0:
            f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    for k in ("fq", "gz", "bgzf.gz"):
        for rep in range(2 if k == "gz" else 5):
            t0 = time.perf_counter()
            rd = FastxReader(paths[k], batch=1 << 19, copy=False, threads=threads)
            tot = sum(len(of) - 1 for _, of in rd)
            rd.close()
            dt = time.perf_counter() - t0
            if tot != n: print("MISMATCH", k, tot, n)
            print("%-8s %6.0f MB: %.3f s  %.2f M reads/s" % (k, os.path.getsize(paths[k]) / 1e6, dt, tot / dt / 1e6))
finally:
    for p in paths.values():
        if os.path.exists(p):
            os.remove(p)
