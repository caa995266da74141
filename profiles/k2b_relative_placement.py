"""k2b_expand differs by 5-8 % between PROCESSES on one box (12.0 / 12.9 ms for threshold union, 6.0 / 6.3 for full intersection: profiles/r6/
k3r_variants_r6.txt, the k2b_expand column of alternating builds), while moving all the result buffers together inside one process changes nothing
(profiles/r5/k2b_placement_r5.txt). Here: one process; a pad of a different size is allocated BETWEEN the result rows (allocated by the first run)
and the colour lists (allocated by the first expand), so that their distance changes.    usage (GPU box): python profiles/k2b_relative_placement.py [tu]"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, fulgor_amd
tu = len(sys.argv) > 1 and sys.argv[1] == "tu"
fg, gen, desc = bench.prepare_workload("s4546syn")
ix = fulgor_amd.Index(fg, device=0)
n = 10000000
b, o = gen.generate(0, n, 150, 42)
reads = ix.upload_reads(b, o)
ix.timing_enable(True)
algo, tau = (fulgor_amd.THRESHOLD_UNION, 0.8) if tu else (fulgor_amd.FULL_INTERSECTION, 0.0)
pads = [0, 4096, 65536, (1 << 20) + 4096, 2 << 20, (2 << 20) + 8192, 64 << 20, (1 << 30) + (1 << 21) + 4096, 3 << 30, (5 << 30) + 12288, 0, 4096]
for trial, pad_bytes in enumerate(pads):
    res = ix.new_result()
    ix.run(reads, res, algo, tau, 0, n)  # rows, counts, offsets are allocated here
    torch.cuda.synchronize()
    pad = torch.empty(pad_bytes, dtype=torch.uint8, device="cuda:0") if pad_bytes else None
    torch.cuda.synchronize()
    prev, rows = {}, []
    ix.timing_reset()
    for i in range(6):
        ix.run(reads, res, algo, tau, 0, n)
        res.expand()  # the first one allocates the colour lists, behind the pad
        cur = {k: v[0] for k, v in ix.timing().items()}
        rows.append({k: cur.get(k, 0.0) - prev.get(k, 0.0) for k in cur})
        prev = cur
    rows = rows[2:]
    print("trial %2d pad %9.4f GB: " % (trial, pad_bytes / 2**30) + "  ".join("%s %.3f (%.3f-%.3f)" % (k, statistics.median(r[k] for r in rows), min(r[k] for r in rows), max(r[k] for r in rows)) for k in rows[0] if k in ("k2_intersect", "k3_union", "k2b_expand")), flush=True)
    res.close()
    del pad
    torch.cuda.empty_cache()
