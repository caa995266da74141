"""HBM-side traffic per launch of every kernel, from a summarize.py summary of separate --pmc passes.

python profiles/traffic_from_pmc.py profiles/r3/<tag>_summary.txt <reads per launch> [algo [workload]] > profiles/traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KB summed over the dispatches in brackets. What the counters mean on gfx950 was
calibrated in round 3 against kernels with known byte counts (profiles/micro/fetch_calibration.hip,
profiles/r3/fetch_calibration.txt):
  * FETCH_SIZE counts 64 bytes per request that leaves the L2, whether the request is for 64 or for 128 bytes: a coalesced
    stream (16 or 4 bytes per lane) reports exactly 1/2 of its bytes, random fetches of 64 / 32 / 16 / 4 bytes report 64 bytes
    each (factor 1.0 of the lines touched), a random 576-byte row (4.5 lines of 128 bytes) reports 0.556 of its bytes.
  * WRITE_SIZE reports the bytes written (1.00 for a coalesced fill, 1.02 for 576-byte nontemporal rows, 1.05 for CSR runs at
    4-byte alignment) and 32 bytes for an isolated 4-byte store.
Hence one factor per kernel, by its dominant fetch pattern (FACTORS below); round 2 doubled every kernel's FETCH_SIZE, which
overstated the lookup kernel's traffic by 2x. Infinity-Cache hits are included: this is traffic below the L2, not DRAM-only."""
import json
import re
import sys

# kernel -> (bench.py timing slot, FETCH_SIZE factor, why)
FACTORS = {
    "k1_lookup": ("k1_lookup", 1.0, "random 64-byte buckets, one per minimizer run (k_gather<4>: 0.999); the 150 read bases per read are a coalesced stream (1.5 of 15 GB) counted at 1/2"),
    "k2r_intersect": ("k2_intersect", 1.8, "random 576-byte rows (k_row576: 0.556)"),
    "k2a_intersect": ("k2_intersect", 1.3, "32-byte descriptor gathers and block bursts of 64-128 bytes (1.0) mixed with 576-byte bitmap rows (1.8): between 1.0 and 1.8"),
    "k3r_union": ("k3_union", 2.0, "row words read 256 bytes per wave-instruction (coalesced: 0.500)"),
    "k3a_union": ("k3_union", 1.3, "as k2a_intersect"),
    "k_generic": ("k2_intersect", 1.0, "32-byte op records (k_gather<2>: 64 bytes reported per 32 fetched)"),
    "k2b_expand": ("k2b_expand", 2.0, "consecutive result rows, coalesced (k_stream4 / k_stream16: 0.500)"),
}

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
    m = re.match(r"^(\S+).*?\[(\d+)\]\s+(FETCH_SIZE|WRITE_SIZE)=([0-9.e+]+)", line)
    if m and sec in ("pmc_fetch", "pmc_write"):
        name, n, ctr, v = m.group(1), int(m.group(2)), m.group(3), float(m.group(4))
        if name in early_exit:
            n -= 1  # that launch moved no data
        vals.setdefault(name, {})[ctr] = v * 1024.0 / n
out = {"source": path, "reads_per_launch": reads, "algo": sys.argv[3] if len(sys.argv) > 3 else "full-intersection",
       "workload": sys.argv[4] if len(sys.argv) > 4 else "s4546syn", "unit": "bytes per launch",
       "method": "factor x FETCH_SIZE + WRITE_SIZE (KB -> bytes), separate rocprofv3 --pmc passes, averaged over the dispatches; "
                 "factor per kernel from profiles/r3/fetch_calibration.txt",
       "kernels": {}}
for k, v in vals.items():
    if k in FACTORS:
        slot, factor, why = FACTORS[k]
        raw = v.get("FETCH_SIZE", 0)
        out["kernels"][slot] = {"kernel": k, "fetch_raw": int(raw), "fetch_factor": factor, "fetch_pattern": why,
                                "fetch": int(factor * raw), "write": int(v.get("WRITE_SIZE", 0)),
                                "total": int(factor * raw + v.get("WRITE_SIZE", 0))}
json.dump(out, sys.stdout, indent=1)
print()
