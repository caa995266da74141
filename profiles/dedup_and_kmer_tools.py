"""Rates of the paths the bench line does not time (round-5 review, item 7): `pseudoalign --deduplicate` on the core-heavy index (where
id lists DO repeat) against the direct path, and the two per-k-mer tools (`kmer-conservation`, `kmer-matches`) through their native line
emitters — the loops of fulgor_amd/cli.py in one process with the index open, input a FASTQ file on tmpfs, output /dev/null.
python profiles/dedup_and_kmer_tools.py [reads for dedup] [reads for kmer-conservation] [reads for kmer-matches]"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import fulgor_amd
from fulgor_amd import driver, synth
from fulgor_amd.index import KmerEmitter
from fulgor_amd.reads import FastxReader, ReadGenerator
n_dedup = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
n_cons = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
n_match = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))


def fastq(path, b, n):
    rec = np.empty((n, 316), dtype=np.uint8)
    ids = np.arange(n, dtype=np.int64)
    rec[:, 0], rec[:, 1], rec[:, 11] = ord("@"), ord("r"), ord("\n")
    for d in range(9):
        rec[:, 2 + d] = ord("0") + (ids // 10 ** (8 - d)) % 10
    rec[:, 12:162] = np.asarray(b[:n * 150]).reshape(n, 150)
    rec[:, 162:165] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, 165:-1] = ord("I")
    rec[:, -1] = ord("\n")
    rec.tofile(path)


def tool_loop(ix, path, tool, batch):
    rd = FastxReader(path, batch=batch, copy=False)
    em = KmerEmitter(ix, tool)
    n = out_bytes = 0
    with open("/dev/null", "wb") as out:
        while True:
            pb, po, cnt = rd.next_raw()
            if cnt == 0:
                break
            pn, pno = rd.names_raw()
            out_bytes += em.write(pb, po, pn, pno, cnt, out.fileno())
            n += cnt
    em.close()
    rd.close()
    return n, out_bytes


path = "/dev/shm/tools_%d.fq" % os.getpid()
try:
    order = [("s4546core (core-heavy profile)", synth.ensure_s4546_core), ("s4546syn", synth.ensure_s4546)]
    if os.environ.get("FULGOR_TOOLS_SWAP"):
        order.reverse()
    for name, ensure in order:
        fg, extra = ensure(os.path.join(ROOT, "data"), g)
        ix = fulgor_amd.Index(fg, device=0)
        gen = ReadGenerator(g, raw_sequences=extra)
        n = max(n_dedup, n_cons, n_match)
        b, o = gen.generate(0, n, 150, 42)
        fastq(path, b, n_dedup)
        print("index %s, %d colours" % (name, ix.num_colors()), flush=True)
        for dedup in (0, 0, 1, 1, 2):  # direct (twice: the first run of the process pins the buffers), device-side grouping, the round-2 host path
            if dedup == 2 and n_dedup > 200000:
                fastq(path, b, 200000)
            t0 = time.perf_counter()
            rd = FastxReader(path, batch=1 << 19, copy=False)
            ix.tune(deduplicate=dedup == 1)
            with open("/dev/null", "wb") as out:
                if dedup == 2:
                    got, mapped = driver.pseudoalign_stream(ix, rd, 0, 0.0, sink=out, fmt="compressed", deduplicate=True)
                else:
                    got, mapped = ix.pseudoalign_stream(rd, out.fileno(), 0, 0.0, 2, 0, True, 0)
            ix.tune(deduplicate=False)
            rd.close()
            dt = time.perf_counter() - t0
            print("  pseudoalign %s: %d reads (%d mapped) in %.3f s = %.2f M reads/s" % (
                ("(direct)", "--deduplicate (device-side grouping)", "--deduplicate (round-2 host path: numpy)")[dedup], got, mapped, dt, got / dt / 1e6), flush=True)
        # how many id lists are distinct (what deduplication can save)
        ido, ids = ix.fetch_color_set_ids_batch(b[:200000 * 150], o[:200001])
        ido = ido.astype(np.int64)
        keys = set(ids[ido[i]:ido[i + 1]].tobytes() for i in range(200000))
        print("  distinct id lists among the first 200000 reads: %d (%.1f %%)" % (len(keys), 100.0 * len(keys) / 200000), flush=True)
        for tool, tname, cnt in ((0, "kmer-conservation", n_cons), (1, "kmer-matches", n_match)):
            fastq(path, b, cnt)
            batch = 1 << 16 if tool == 0 else max(256, min(1 << 16, (1 << 26) // ix.num_colors()))
            for rep in range(2):
                t0 = time.perf_counter()
                got, ob = tool_loop(ix, path, tool, batch)
                dt = time.perf_counter() - t0
                print("  %-18s %d records in %.3f s = %.3f M records/s, %.2f GB of text (%.0f bytes per line, %.2f GB/s), batches of %d" % (
                    tname, got, dt, got / dt / 1e6, ob / 1e9, ob / max(1, got), ob / dt / 1e9, batch), flush=True)
        ix.close()
finally:
    if os.path.exists(path):
        os.remove(path)
