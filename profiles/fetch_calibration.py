#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE as rocprofv3 reports them on gfx950 against the known byte counts of profiles/micro/fetch_calibration.hip.
usage: python profiles/fetch_calibration.py <known.txt> <dir with the --pmc FETCH_SIZE pass> <dir with the --pmc WRITE_SIZE pass>"""
import csv
import glob
import os
import sys

known = {}
for line in open(sys.argv[1]):
    t = line.split()
    if len(t) == 5 and t[1] in ("read", "write"):
        known[t[0]] = (t[1], int(t[2]), int(t[3]), int(t[4]))
got = {}
for d, ctr in ((sys.argv[2], "FETCH_SIZE"), (sys.argv[3], "WRITE_SIZE")):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                got.setdefault(name, {}).setdefault(ctr, 0.0)
                got[name][ctr] += float(r["Counter_Value"]) * 1024.0  # reported in KB
print("%-14s %-5s %14s %14s %14s | %14s %8s %8s %8s" % ("kernel", "side", "useful B", "64-B lines B", "128-B lines B", "counter B", "/useful", "/64B", "/128B"))
for k, (side, useful, l64, l128) in known.items():
    v = got.get(k, {}).get("FETCH_SIZE" if side == "read" else "WRITE_SIZE")
    if v is None:
        print("%-14s %-5s %14d %14d %14d | (no counter row)" % (k, side, useful, l64, l128))
        continue
    print("%-14s %-5s %14d %14d %14d | %14d %8.3f %8.3f %8.3f" % (k, side, useful, l64, l128, v, v / useful, v / l64, v / l128))
