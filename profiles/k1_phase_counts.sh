#!/bin/bash
# instruction counts of the lookup kernel up to the end of phase A / B / C (knock-out builds -DFG_K1_STOP=1|2|3) and of the
# whole kernel: rocprofv3 --pmc on profiles/k1_variant_time.py. usage: bash profiles/k1_phase_counts.sh <lib.so> ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  rm -rf /tmp/k1pc
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/k1pc -o x -- python $R/profiles/k1_variant_time.py 2500000 $R/$so > /tmp/k1pc.log 2>&1
  python - "$so" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc, n = defaultdict(float), set()
for f in glob.glob("/tmp/k1pc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k1_lookup" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
reads = 2500000 * max(1, len(n))
print(sys.argv[1], "launches", len(n), " ".join("%s/read=%.1f" % (k, v / reads) for k, v in sorted(acc.items()) if k != "SQ_WAVES"))
PY
  grep "\.so" /tmp/k1pc.log | tail -1
done
