"""Threshold union on reads of 300 / 400 bases (16-bit score counters: k3r_union<16>) and of 40000 k-mers (32-bit: k3r_union<32>) on the
bench index (the 40000-k-mer case went when the generator stopped finding windows that long): kernel times of alternative builds. python profiles/k3r_long_reads_time.py <lib.so>[,<lib.so>...]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "," in sys.argv[1]:
    for so in sys.argv[1].split(","):
        subprocess.run([sys.executable, __file__, so])
    raise SystemExit(0)
so = os.path.abspath(sys.argv[1])
from fulgor_amd import _build
_build.LIB_GPU = so
import numpy as np
import fulgor_amd
from fulgor_amd import synth
from fulgor_amd.reads import ReadGenerator
g = sorted(glob.glob(os.path.join(ROOT, "tests", "data", "salmonella_10", "*.fasta.gz")))
fg, extra = synth.ensure_s4546(os.path.join(ROOT, "data"), g)
gen = ReadGenerator(g, raw_sequences=extra)
ix = fulgor_amd.Index(fg, device=0)
for n, length in ((1_000_000, 200), (1_000_000, 300), (1_000_000, 400), (1_000_000, 540)):
    b, o = gen.generate(0, n, length, 42)
    reads = ix.upload_reads(b, o)
    res = ix.new_result()
    for _ in range(2):
        ix.run(reads, res, fulgor_amd.THRESHOLD_UNION, 0.8, 0, n)
    ix.timing_enable(True)
    ix.timing_reset()
    for _ in range(4):
        ix.run(reads, res, fulgor_amd.THRESHOLD_UNION, 0.8, 0, n)
    tm = ix.timing()
    ix.timing_enable(False)
    print("%s  %d reads of %d bases: %s" % (os.path.basename(so), n, length, ", ".join("%s %.3f ms" % (k_, v[0] / v[1]) for k_, v in tm.items() if v[1])), flush=True)
    res.close(); reads.close()
