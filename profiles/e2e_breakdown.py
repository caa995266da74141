"""Where the PCIe-inclusive pass spends its time (bench.py's end_to_end leg): upload of the reads from pageable and
from pinned host memory, kernels, device formatter + D2H. usage: python profiles/e2e_breakdown.py [reads]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, fulgor_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
fg, gen, desc = bench.prepare_workload("s4546syn", 0)
ix = fulgor_amd.Index(fg, device=0)
bases, offs = gen.generate(0, n, 150, 42)
pb = torch.empty(len(bases), dtype=torch.uint8, pin_memory=True)
po = torch.empty(len(offs), dtype=torch.int64, pin_memory=True)
pb.numpy()[:] = np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else bases
po.numpy()[:] = offs.astype(np.int64)
res = ix.new_result()
for name, b, o in (("pageable", bases, offs), ("pinned", pb.numpy(), po.numpy().astype(np.uint64, copy=False))):
    for fmt, fname in ((0, "ascii"), (2, "compressed")):
        best = None
        for _ in range(3):
            t0 = time.perf_counter(); rd = ix.upload_reads(b, o)
            t1 = time.perf_counter(); ix.run(rd, res, fulgor_amd.FULL_INTERSECTION, 0.0)
            t2 = time.perf_counter(); text = res.format_view(fmt, 0)
            t3 = time.perf_counter(); rd.close()
            cur = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
            best = cur if best is None or cur[0] < best[0] else best
        print("%-9s %-10s total %7.2f ms = upload %6.2f + kernels %6.2f + format/D2H %6.2f  (%5.1f M reads/s, %d output bytes)" %
              (name, fname, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3, n / best[0] / 1e6, len(text)))
