"""Device-side formatter throughput: one 2^20-read full-intersection pass on the bench index, then
fgpu_result_format (kernels + D2H of the text) against download + host formatter. python profiles/format_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, fulgor_amd
from fulgor_amd.driver import Formatter

fg, gen, desc = bench.prepare_workload("s4546syn", 0)
ix = fulgor_amd.Index(fg, device=0)
b, o = gen.generate(0, 1 << 20, 150, 42)
rd, res = ix.upload_reads(b, o), ix.new_result()
ix.run(rd, res, fulgor_amd.FULL_INTERSECTION)
n, total, mapped = res.sizes()
ix.timing_enable(True)
for name, code in (("ascii", 0), ("binary", 1), ("compressed", 2)):
    res.format_view(code, 0)
    ix.timing_reset()
    t0 = time.perf_counter(); out = res.format_view(code, 0); t1 = time.perf_counter()
    kms = ix.timing()["k_format"][0]
    if code == 2:
        from oracle.pyoracle import parse_compressed
        offs, cols = res.download()
        t4 = time.perf_counter(); ref = f.add(0, offs, cols) + f.finish() if False else None; t5 = time.perf_counter()
        hf = Formatter(name, ix.num_colors())
        t4 = time.perf_counter(); ref = hf.add(0, offs, cols) + hf.finish(); t5 = time.perf_counter()
        print("compressed: %d reads -> %.3f GB (host formatter: %.3f GB); format kernels %.2f ms; with D2H into the pinned buffer %.1f ms; "
              "host formatter (1 thread) %.0f ms" % (n, len(out) / 1e9, len(ref) / 1e9, kms, (t1 - t0) * 1e3, (t5 - t4) * 1e3))
        continue
    t2 = time.perf_counter(); offs, cols = res.download(); t3 = time.perf_counter()
    f = Formatter(name, ix.num_colors())
    t4 = time.perf_counter(); ref = f.add(0, offs, cols); t5 = time.perf_counter()
    assert bytes(out) == bytes(ref)
    print("%s: %d reads, %d colours -> %.2f GB; format kernels %.2f ms (%.0f GB/s of text); with D2H into the pinned buffer %.0f ms; "
          "CSR download %.0f ms + host formatter (1 thread) %.0f ms"
          % (name, n, total, len(out) / 1e9, kms, len(out) / 1e6 / kms, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t5 - t4) * 1e3))
