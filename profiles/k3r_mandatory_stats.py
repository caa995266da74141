"""Threshold union, which lists are MANDATORY: a colour that misses list l scores at most P - m_l, so a list with
m_l > P - min_score must contain every result colour. How many lists of a read are left once those are taken out?
(decides whether the multiplexer tree of k3r_union can serve reads of more than 6 lists). python profiles/k3r_mandatory_stats.py [reads] [tau]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, fulgor_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
tau = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
fg, gen, desc = bench.prepare_workload("s4546syn")
ix = fulgor_amd.Index(fg, device=0)
b, o = gen.generate(0, n, 150, 42)
ko, ki = ix.kmer_color_set_ids_batch(b, o)
ko = ko.astype(np.int64)
nl_all, free_all, P_all, slack_all = [], [], [], []
free_mu = []  # multiplicities of the not-mandatory lists of the reads that have more than six of them
for r in range(n):
    ids = ki[ko[r]:ko[r + 1]]
    ids = ids[ids != 0xFFFFFFFF]
    if len(ids) == 0:
        nl_all.append(0); free_all.append(0); P_all.append(0); slack_all.append(0); continue
    _, m = np.unique(ids, return_counts=True)
    P = int(m.sum()); ms = int(P * tau); slack = P - ms
    nl_all.append(len(m)); free_all.append(int((m <= slack).sum())); P_all.append(P); slack_all.append(slack)
    if (m <= slack).sum() > 6:
        free_mu.extend(m[m <= slack].tolist())
nl, fr = np.array(nl_all), np.array(free_all)
print("workload:", desc)
print("reads %d tau %.2f; lists per read mean %.2f; reads with > 6 lists %.1f%%" % (n, tau, nl.mean(), 100 * (nl > 6).mean()))
big = nl > 6
print("of those: lists mean %.2f, not mandatory mean %.2f; with <= 6 not mandatory %.1f%%, <= 5: %.1f%%, <= 4: %.1f%%" % (
    nl[big].mean(), fr[big].mean(), 100 * (fr[big] <= 6).mean(), 100 * (fr[big] <= 5).mean(), 100 * (fr[big] <= 4).mean()))
print("all reads with lists: not mandatory mean %.2f; histogram of not-mandatory lists:" % fr[nl > 0].mean(),
      " ".join("%d:%.1f%%" % (i, 100.0 * c / (nl > 0).sum()) for i, c in enumerate(np.bincount(fr[nl > 0])) if c))
print("histogram of lists:", " ".join("%d:%.1f%%" % (i, 100.0 * c / (nl > 0).sum()) for i, c in enumerate(np.bincount(nl[nl > 0])) if c))
sl = np.array(slack_all)
over = fr > 6
print("reads whose not-mandatory lists exceed 6 (the byte-counter path): %.1f%% of the reads with lists; they hold %.1f%% of all (read, list) pairs; their not-mandatory lists: %s" % (
    100.0 * over.sum() / (nl > 0).sum(), 100.0 * nl[over].sum() / max(1, nl.sum()),
    " ".join("%d:%.1f%%" % (i, 100.0 * c / max(1, over.sum())) for i, c in enumerate(np.bincount(fr[over])) if c and i <= 24)))
if over.any():
    bits = np.ceil(np.log2(sl[over] + 2)).astype(int)
    print("slack P - min_score of those reads: mean %.1f, max %d; bits of a saturating deficit counter (ceil log2(slack + 2)): %s" % (
        sl[over].mean(), sl[over].max(), " ".join("%d:%.1f%%" % (i, 100.0 * c / over.sum()) for i, c in enumerate(np.bincount(bits)) if c)))
if free_mu:
    fm = np.array(free_mu)
    print("multiplicities of those reads' not-mandatory lists: mean %.1f; bits: %s; values 1..8: %s" % (
        fm.mean(), " ".join("%d:%.1f%%" % (i, 100.0 * c / len(fm)) for i, c in enumerate(np.bincount(np.ceil(np.log2(fm + 1)).astype(int))) if c),
        " ".join("%d:%.1f%%" % (i, 100.0 * (fm == i).mean()) for i in range(1, 9))))
