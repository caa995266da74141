"""HBM-side traffic per launch of every kernel, from a summarize.py summary of separate --pmc passes.

python profiles/traffic_from_pmc.py profiles/r1/<tag>_s4546syn_1M_summary.txt <reads per launch> > profiles/traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KB summed over the dispatches in brackets. On gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes (MI355X_MICROARCH.md, "HBM"), hence the factor 2; WRITE_SIZE is used as reported
(both were checked on this pipeline against kernels with known byte counts: k2b_expand's output, k2a's bitmaps).
Infinity-Cache hits are included: this is traffic below the L2, not DRAM-only traffic."""
import json, re, sys

path, reads = sys.argv[1], int(sys.argv[2])
sec = None
vals = {}
early_exit = set()  # kernels with one launch that returned at once (k2b_expand's device-side capacity check on the first pass)
for line in open(path):
    if line.startswith("== "):
        sec = line.split()[1].rstrip(":") if not line.startswith("== kernel trace") else "trace"
        continue
    if sec == "trace":
        f = line.split()
        if len(f) == 5 and f[1].isdigit() and float(f[3]) < 0.02 * float(f[4]):
            early_exit.add(f[0])
        continue
    m = re.match(r"^(\S+)\s+\[(\d+)\]\s+(FETCH_SIZE|WRITE_SIZE)=([0-9.e+]+)", line)
    if m and sec in ("pmc_fetch", "pmc_write"):
        name, n, ctr, v = m.group(1), int(m.group(2)), m.group(3), float(m.group(4))
        if name in early_exit:
            n -= 1  # that launch moved no data
        vals.setdefault(name, {})[ctr] = v * 1024.0 / n
out = {"source": path, "reads_per_launch": reads, "algo": sys.argv[3] if len(sys.argv) > 3 else "full-intersection",
       "unit": "bytes per launch",
       "method": "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), separate rocprofv3 --pmc passes, averaged over the dispatches",
       "kernels": {}}
for k, v in vals.items():
    if k.startswith("k"):
        out["kernels"][k] = {"fetch": int(2 * v.get("FETCH_SIZE", 0)), "write": int(v.get("WRITE_SIZE", 0)),
                             "total": int(2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0))}
json.dump(out, sys.stdout, indent=1)
print()
