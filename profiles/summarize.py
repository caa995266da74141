#!/usr/bin/env python3
"""Summarises a gpurun_out/prof_<tag>/ directory written by profiles/collect.sh into one text file:
per-kernel durations from the kernel trace, and per-kernel sums of every PMC counter."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("k1_lookup", "k2a_intersect", "k2r_intersect", "k3a_union", "k3r_union", "k2b_expand", "k_hits", "scan_block_sums", "scan_top",
              "scan_apply", "k_account", "k_add_totals", "k2_fused", "k_order_keys", "k_order_scatter", "k_rows_build", "k_generic", "k_desc"):
        if k in name:
            return k
    return name[:60]


def main(d):
    out = []
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
        out.append("== kernel stats (%s)" % os.path.relpath(f, d))
        out.append(open(f).read())
    for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_trace.csv"), recursive=True):
        dur = defaultdict(list)
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        out.append("== kernel trace durations (us): name count avg min max")
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            out.append("%-18s %5d %12.1f %12.1f %12.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
    for p in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        if not os.path.isdir(p):
            continue
        for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
            acc = defaultdict(lambda: defaultdict(float))
            cnt = defaultdict(int)
            seen = set()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (r["Dispatch_Id"], k)
                if key not in seen:
                    seen.add(key)
                    cnt[k] += 1
            out.append("== %s: per-kernel counter sums over all dispatches (dispatch count in brackets)" % os.path.basename(p))
            for k in acc:
                out.append("%-18s [%d] " % (k, cnt[k]) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(acc[k].items())))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
