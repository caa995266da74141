import os, struct, sys, time, zlib
sys.path.insert(0, "/root/repo")
import numpy as np
from fulgor_amd.reads import FastxReader
n = 3_000_000
rng = np.random.default_rng(3)
rec = np.empty((n, 316), dtype=np.uint8)
rec[:, 0], rec[:, 1], rec[:, 11] = ord("@"), ord("r"), ord("\n")
rec[:, 2:11] = ord("0")
rec[:, 12:162] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 150))]
rec[:, 162:165] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 165:-1] = rng.integers(35, 74, size=(n, 150), dtype=np.uint8)
rec[:, -1] = ord("\n")
plain = rec.tobytes(); del rec
p = "/dev/shm/x.bgzf.gz"
with open(p, "wb") as f:
    for at in range(0, len(plain), 65280):
        blk = plain[at:at + 65280]
        co = zlib.compressobj(1, zlib.DEFLATED, -15); cd = co.compress(blk) + co.flush()
        f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd + struct.pack("<II", zlib.crc32(blk) & 0xFFFFFFFF, len(blk)))
    f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
for rep in range(3):
    t0 = time.perf_counter(); rd = FastxReader(p, batch=1 << 19, copy=False, threads=int(sys.argv[1]) if len(sys.argv) > 1 else 0); t1 = time.perf_counter()
    it = iter(rd); first = next(it); t2 = time.perf_counter()
    tot = len(first[1]) - 1 + sum(len(of) - 1 for _, of in it); t3 = time.perf_counter()
    rd.close(); t4 = time.perf_counter()
    print("open %.3f first batch %.3f rest %.3f close %.3f total %.3f reads %d" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, tot))
os.remove(p)
