"""Shape of the colour-set work in the bench workload (CPU only; uses the oracle for the k-mer lookups).

python profiles/workload_stats.py [--reads 100000]
Prints: lists per read, encoding mix, codes per list, 16-code segments per read — the numbers behind the
lane-utilisation discussion in DESIGN.md."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fulgor_amd  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402


def delta_at(words, pos):
    def bits(p, n):
        v = 0
        for i in range(n):
            v |= ((int(words[(p + i) >> 6]) >> ((p + i) & 63)) & 1) << i
        return v
    z = 0
    while bits(pos + z, 1) == 0:
        z += 1
    ln = (bits(pos + z + 1, z) | (1 << z)) - 1
    body = bits(pos + 2 * z + 1, ln)
    return (body | (1 << ln)) - 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--workload", default="s4546syn")
    a = ap.parse_args()
    fg, gen, desc = bench.prepare_workload(a.workload, 0)
    ix = fulgor_amd.Index(fg, device=-1)
    ex = ix.export()
    n = ix.num_colors()
    sparse_thr, dense_thr = int(ex["thresholds"][1]), int(ex["thresholds"][2])
    orc = pyoracle.OracleIndex.from_export(ex)
    bases, offs = gen.generate(0, a.reads, 150, 42)
    io, ids = orc.fetch_color_set_ids(bases, offs)
    io = np.asarray(io, dtype=np.int64)
    per_read = np.diff(io)
    uniq = np.unique(ids)
    size = {int(i): delta_at(ex["color_words"], int(ex["color_offsets"][i])) for i in uniq}
    sz = np.array([size[int(i)] for i in ids])
    bits = np.array([int(ex["color_offsets"][int(i) + 1] - ex["color_offsets"][int(i)]) for i in ids])
    kind = np.where(sz < sparse_thr, 0, np.where(sz < dense_thr, 1, 2))
    ncodes = np.where(kind == 0, sz, np.where(kind == 1, 0, n - sz))
    nseg = (ncodes + 15) // 16
    seg_per_read = np.add.reduceat(nseg, io[:-1][per_read > 0]) if len(ids) else np.array([])
    print("workload:", desc)
    print("reads %d; lists/read mean %.2f  p50 %d p90 %d p99 %d max %d; reads with 0 lists %.1f%%" % (
        a.reads, per_read.mean(), *np.percentile(per_read, [50, 90, 99]).astype(int), per_read.max(), 100 * (per_read == 0).mean()))
    for k, name in enumerate(["delta-gaps", "bitmap", "complement"]):
        m = kind == k
        if m.any():
            print("  %-10s %5.1f%% of lists; codes/list mean %.0f p50 %.0f p90 %.0f; bits/list mean %.0f" % (
                name, 100 * m.mean(), ncodes[m].mean(), *np.percentile(ncodes[m], [50, 90]), bits[m].mean()))
    print("16-code segments per read: mean %.1f p50 %d p90 %d p99 %d  -> lanes busy in the decode loop %.0f%%" % (
        seg_per_read.mean(), *np.percentile(seg_per_read, [50, 90, 99]).astype(int),
        100 * (seg_per_read / (np.ceil(seg_per_read / 64) * 64 + 1e-9)).mean()))
    print("compressed list bytes per read: %.0f" % (bits.sum() / 8 / a.reads))


if __name__ == "__main__":
    main()
