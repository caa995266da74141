"""Dry run of the real-index path at full size (round-5 review, item 4): the bench index (4546 colours, 0.85 M colour sets, 0.9 G
integers) is written as the reference's dump files (`fulgor dump`, src/index.cpp:59-120), ingested again through the path a real
salmonella_4546 dump takes (FULGOR_S4546_DUMP -> load_dump: colour sets parsed and encoded on all threads), and saved: the container
must be the original byte for byte. With --bench, bench.py then runs on the synthetic index and through the FULGOR_S4546_DUMP hook on
the dump (with the synthetic workload's reads, so that the two lines compare number for number).
python profiles/dump_roundtrip.py [--bench]"""
import glob, hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fulgor_amd
from fulgor_amd import synth
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, _ = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
tmp = "/dev/shm/fulgor_dump_%d" % os.getpid()
os.makedirs(tmp, exist_ok=True)
base = os.path.join(tmp, "salmonella_4546_synth")
try:
    t0 = time.perf_counter()
    ix = fulgor_amd.Index(fg, device=-1)
    t1 = time.perf_counter()
    ix.dump(base)
    t2 = time.perf_counter()
    info = (ix.num_colors(), ix.num_color_sets(), ix.num_unitigs(), ix.num_kmers())
    ix.close()
    sizes = {s: os.path.getsize(base + s) for s in (".color_sets.txt", ".unitigs.fa", ".metadata.txt", ".filenames.txt")}
    print("index %s: %d colours, %d colour sets, %d unitigs, %d k-mers" % ((os.path.basename(fg),) + info))
    print("host-only open of the container %.2f s; dump written in %.2f s: %s" % (t1 - t0, t2 - t1, ", ".join("%s %.3f GB" % (k, v / 1e9) for k, v in sizes.items())))
    nints = 0
    with open(base + ".color_sets.txt", "rb") as f:
        while True:
            blk = f.read(1 << 26)
            if not blk:
                break
            nints += blk.count(b" ")
    print("colour sets file: %.3f G integers" % (nints / 1e9))
    t3 = time.perf_counter()
    env = dict(os.environ, FULGOR_VERBOSE_LOAD="1")
    iy = fulgor_amd.Index(base, device=-1)
    t4 = time.perf_counter()
    again = os.path.join(tmp, "again.fgidx")
    iy.save(again)
    t5 = time.perf_counter()
    iy.close()
    print("ingest of the dump (load_dump on %d hardware threads: colour sets, packed blocks, unitigs, dictionary): %.2f s; container saved in %.2f s" % (os.cpu_count(), t4 - t3, t5 - t4))

    def sha(p):
        h = hashlib.sha256()
        with open(p, "rb") as f:
            while True:
                b = f.read(1 << 26)
                if not b:
                    break
                h.update(b)
        return h.hexdigest()
    a, b = sha(fg), sha(again)
    print("container of the re-ingested dump: %d bytes, sha256 %s; original: %d bytes, sha256 %s: %s" % (os.path.getsize(again), b[:16], os.path.getsize(fg), a[:16], "IDENTICAL" if a == b else "DIFFERENT"))
    if a != b:
        raise SystemExit(1)
    if "--bench" in sys.argv:
        lines = {}
        for tag, extra_env in (("synthetic", {}), ("dump", {"FULGOR_S4546_DUMP": base, "FULGOR_S4546_DUMP_READS": "synthetic"})):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-secondary"],
                               env=dict(os.environ, **extra_env), capture_output=True, text=True, timeout=3000)
            assert r.returncode == 0, r.stderr[-3000:]
            lines[tag] = json.loads(r.stdout.strip().splitlines()[-1])
            l = lines[tag]
            print("bench.py on the %s index: %.1f M reads/s, %.3f ms per step, kernels %s, data=%s, %.1f colours per read, mapped %.3f" % (
                tag, l["value"] / 1e6, l["ms_per_step"], l["kernels_ms"], l["data"], l["config"]["avg_colours_per_read"], l["config"]["mapped_fraction"]))
        d = lines["dump"]["value"] / lines["synthetic"]["value"] - 1
        print("dump line against synthetic line: %+.2f %% (same reads: FULGOR_S4546_DUMP_READS=synthetic; a real dump draws its reads from its own unitigs)" % (d * 100))
finally:
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    # (bench.py's cache of the ingested dump)
    for p in glob.glob(os.path.join(ROOT, "data", "salmonella_4546_synth.v9.fgidx*")):
        os.remove(p)
