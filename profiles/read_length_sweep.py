"""Kernel times by read length on the bench index (full intersection, 2 M reads each): the lookup kernel has instantiations for units
of up to 128 / 256 / 512 k-mers (158 / 286 / 542 bases at k = 31). python profiles/read_length_sweep.py"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import ReadGenerator
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
gen = ReadGenerator(g, raw_sequences=extra)
ix = fulgor_amd.Index(fg, device=0)
n = 2_000_000
for length in [int(x) for x in os.environ.get("FULGOR_SWEEP_LENGTHS", "75,100,125,150,158,159,180,200,222,223,250,286,287,300,400,542").split(",")]:
    b, o = gen.generate(0, n, length, 42)
    reads = ix.upload_reads(b, o)
    res = ix.new_result()
    for _ in range(2):
        ix.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, 0, n)
    ix.timing_enable(True)
    ix.timing_reset()
    for _ in range(4):
        ix.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, 0, n)
    tm = ix.timing()
    ix.timing_enable(False)
    k1 = tm["k1_lookup"][0] / tm["k1_lookup"][1]
    tot = sum(v[0] / v[1] for v in tm.values() if v[1])
    kmers = n * (length - 30)
    print("%4d bases: k1_lookup %.3f ms = %.0f G k-mers/s; all kernels of the pass (no u32 lists) %.3f ms = %.0f M reads/s" % (length, k1, kmers / k1 / 1e6, tot, n / tot / 1e3), flush=True)
    res.close(); reads.close()
