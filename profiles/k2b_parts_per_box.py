"""Which part of k2b_expand carries the box-to-box difference? On ONE box: the product build and the three knock-outs (build/variants/k2bko{1,2,3}.so:
no hit-counter adds / no stage scatter / no colour stores; results are wrong, times are the point), 8 passes each in a process of its own.
usage (GPU box): python profiles/k2b_parts_per_box.py >> gpurun_out/k2b_parts.txt"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, statistics, sys
sys.path.insert(0, %r)
import bench, fulgor_amd
fg, gen, desc = bench.prepare_workload("s4546syn")
ix = fulgor_amd.Index(fg, device=0)
b, o = gen.generate(0, 10000000, 150, 42)
reads = ix.upload_reads(b, o)
res = ix.new_result()
ix.timing_enable(True)
prev, rows = {}, []
for i in range(9):
    ix.run(reads, res, 0, 0.0, 0, 10000000)
    res.expand()
    cur = {k: v[0] for k, v in ix.timing().items()}
    rows.append({k: cur[k] - prev.get(k, 0.0) for k in ("k1_lookup", "k2_intersect", "k2b_expand")})
    prev = cur
rows = rows[2:]
print(" ".join("%%s %%.3f" %% (k, statistics.median(r[k] for r in rows)) for k in ("k1_lookup", "k2_intersect", "k2b_expand")))
''' % ROOT
uid = subprocess.run("rocm-smi --showuniqueid 2>/dev/null | grep -o '0x[0-9a-f]*'", shell=True, capture_output=True, text=True).stdout.strip()
out = ["gpu %s" % uid]
for v in ("product", "k2bko1", "k2bko2", "k2bko3"):
    env = dict(os.environ)
    if v != "product":
        env["FULGOR_LIB_GPU"] = os.path.join(ROOT, "build", "variants", v + ".so")
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("k1_lookup")]
    out.append("%s: %s" % (v, line[-1] if line else "failed: " + r.stderr[-300:]))
print(" | ".join(out), flush=True)
