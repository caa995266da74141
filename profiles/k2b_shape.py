"""Shape of the RESULT bitmaps of the bench workload (runs on the GPU): how many colours per read, how they spread
over the 32-colour words one lane of k2b_expand owns, and what the per-bit loop pays for the imbalance.

python profiles/k2b_shape.py [--reads 200000] [--algo fi|tu]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fulgor_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=200000)
    ap.add_argument("--workload", default="s4546syn")
    ap.add_argument("--algo", default="fi")
    a = ap.parse_args()
    fg, gen, desc = bench.prepare_workload(a.workload, 0)
    ix = fulgor_amd.Index(fg, device=0)
    n = ix.num_colors()
    W = (n + 31) // 32
    bases, offs = gen.generate(0, a.reads, 150, 42)
    if a.algo == "fi":
        off, col = ix.pseudoalign_full_intersection_batch(bases, offs)
    else:
        off, col = ix.pseudoalign_threshold_union_batch(bases, offs, 0.8)
    off = np.asarray(off, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    sizes = np.diff(off)
    read_of = np.repeat(np.arange(a.reads), sizes)
    pc = np.zeros((a.reads, W), dtype=np.int32)
    np.add.at(pc, (read_of, col >> 5), 1)
    print("workload:", desc, "algo", a.algo)
    print("colours per read: mean %.1f p50 %d p90 %d p99 %d; empty %.1f%%" % (
        sizes.mean(), *np.percentile(sizes, [50, 90, 99]).astype(int), 100 * (sizes == 0).mean()))
    nz = sizes > 0
    rounds = (W + 63) // 64
    pad = np.zeros((a.reads, rounds * 64), dtype=np.int32)
    pad[:, :W] = pc
    mx = pad.reshape(a.reads, rounds, 64).max(axis=2)
    print("words per read: %d in %d rounds; per-bit loop iterations per read (sum of round maxima): mean %.1f (non-empty reads %.1f)" % (
        W, rounds, mx.sum(axis=1).mean(), mx.sum(axis=1)[nz].mean()))
    print("balanced bound (colours / 64 lanes): mean %.1f" % (sizes[nz] / 64.0).mean())
    for thr in (4, 8, 12, 16, 24, 31):
        print("  words with more than %2d colours per read: mean %.1f" % (thr, (pc[nz] > thr).sum(axis=1).mean()))
    h = np.bincount(pc[nz].ravel(), minlength=33)
    print("word popcount histogram (non-empty reads):", " ".join("%d:%.1f%%" % (i, 100.0 * h[i] / h.sum()) for i in range(33) if h[i]))
    # capped variants: first CAP bits by the per-lane loop, the rest of a word by a cooperative pass (32 lanes per word)
    for cap in (4, 6, 8, 12):
        it = np.minimum(mx, cap).sum(axis=1)[nz].mean()
        dense = (pc[nz] > cap).sum(axis=1).mean()
        print("  cap %2d: per-lane iterations %.1f + dense words %.1f (two per cooperative step: %.1f steps)" % (cap, it, dense, dense / 2))


if __name__ == "__main__":
    main()
