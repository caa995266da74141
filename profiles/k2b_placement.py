"""Does the time of k2b_expand depend on WHERE its buffers lie? Ten different GPUs run it in 5.98-6.12 ms from a fresh process that allocates in
the same order (profiles/k2b_box_spread.py), while bench.py lines of the same build show 5.57 / 6.00 / 6.33. Here: one process, one GPU; before
every trial a pad of a different size is allocated (and kept) so that the result buffers (rows 5.8 GB, colours 24 GB) land elsewhere.
usage (GPU box): python profiles/k2b_placement.py"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, fulgor_amd
fg, gen, desc = bench.prepare_workload("s4546syn")
ix = fulgor_amd.Index(fg, device=0)
b, o = gen.generate(0, 10000000, 150, 42)
reads = ix.upload_reads(b, o)
ix.timing_enable(True)
pads = [0, 1 << 20, (100 << 20) + 4096, 1 << 30, (3 << 30) + 65536, (7 << 30) + (1 << 21), (20 << 30) + 12288, 50 << 30, (90 << 30) + 4096, 0, 1 << 30]
for trial, pad_bytes in enumerate(pads):
    pad = torch.empty(pad_bytes, dtype=torch.uint8, device="cuda:0") if pad_bytes else None
    torch.cuda.synchronize()
    res = ix.new_result()
    prev, rows = {}, []
    for i in range(8):
        ix.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, 0, 10000000)
        res.expand()
        cur = {k: v[0] for k, v in ix.timing().items()}
        rows.append({k: cur[k] - prev.get(k, 0.0) for k in ("k1_lookup", "k2_intersect", "k2b_expand")})
        prev = cur
    rows = rows[2:]
    print("trial %2d pad %6.2f GB: " % (trial, pad_bytes / 2**30) + "  ".join("%s %.3f (%.3f-%.3f)" % (k, statistics.median(r[k] for r in rows), min(r[k] for r in rows), max(r[k] for r in rows)) for k in ("k1_lookup", "k2_intersect", "k2b_expand")), flush=True)
    ix.timing_reset()
    res.close()
    del pad
    torch.cuda.empty_cache()
