import os, sys, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, fulgor_amd
from fulgor_amd import pack_reads
from fulgor_amd.driver import Formatter
import test_gpu_parity as T
n = 12001
rng = np.random.default_rng(n)
base = os.path.join(tempfile.mkdtemp(), "mid")
unitigs, nsets = T._write_wide_dump(base, rng, n=n)
ix = fulgor_amd.Index(base, device=0)
reads = []
for _ in range(600):
    reads.append("".join(u[s:s + 60] for u, s in ((unitigs[rng.integers(len(unitigs))], rng.integers(0, 140)) for _ in range(rng.integers(1, 4)))))
reads += [u[:150] for u in unitigs]
b, o = pack_reads(reads)
rd, res = ix.upload_reads(b, o), ix.new_result()
ix.run(rd, res, fulgor_amd.FULL_INTERSECTION)
go, gc = res.download()
want = bytes(Formatter("ascii", n).add(3, go, gc))
for step in ("ascii", "compressed", "ascii", "ascii"):
    got = bytes(res.format_view(2 if step == "compressed" else 0, 3))
    if step == "ascii":
        same = got == want
        first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), None)
        print(step, "len", len(got), "expected", len(want), "equal", same, "first difference at", first, "| got is a suffix-shift of expected at", want.find(got[:64]))
    else:
        print(step, len(got))
