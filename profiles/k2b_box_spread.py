"""Why does k2b_expand differ by 10 % from box to box (round-4 review, item 6)? One record per box: which GPU (unique id, power cap), clocks /
power / temperature sampled from sysfs while 20 passes of the bench workload run (lookup, intersection, expansion timed by HIP events),
then the write-only microbenchmarks (profiles/micro/hbm_rates, store_patterns) on the same box.
usage (GPU box): python profiles/k2b_box_spread.py >> gpurun_out/k2b_box_spread.txt"""
import glob, os, statistics, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, fulgor_amd


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=120).stdout
    except Exception as e:  # noqa: BLE001
        return "(%s)" % e


def active_card():
    """the drm card of the GPU this process uses: the one whose VRAM use grows when the index goes up"""
    cards = [c for c in glob.glob("/sys/class/drm/card*/device") if os.path.exists(c + "/mem_info_vram_used")]
    return cards


def read(p):
    try:
        return open(p).read().strip()
    except OSError:
        return ""


cards = active_card()
before = {c: int(read(c + "/mem_info_vram_used") or 0) for c in cards}
fg, gen, desc = bench.prepare_workload("s4546syn")
ix = fulgor_amd.Index(fg, device=0)
b, o = gen.generate(0, 10000000, 150, 42)
reads = ix.upload_reads(b, o)
card = max(cards, key=lambda c: int(read(c + "/mem_info_vram_used") or 0) - before[c]) if cards else None
hw = (glob.glob(card + "/hwmon/hwmon*") or [None])[0] if card else None
print("==== box %s  %s" % (os.uname().nodename, time.strftime("%Y-%m-%d %H:%M:%S")))
print(sh("rocm-smi --showuniqueid --showserial --showmaxpower --showperflevel --showmemvendor 2>/dev/null | grep -v '^=\\|^$'").strip())
if card:
    print("card %s: pcie %s, numa node %s, power cap %s uW, vram %s" % (card, os.path.basename(os.path.realpath(card)), read(card + "/numa_node"),
                                                                         read(hw + "/power1_cap") if hw else "?", read(card + "/mem_info_vram_total")))
samples = []
stop = False


def sampler():
    while not stop:
        s = {"t": time.perf_counter()}
        if card:
            for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
                cur = [l for l in read(card + "/" + name).splitlines() if l.endswith("*")]
                s[name] = cur[0].split(":")[1].strip(" *") if cur else ""
        if hw:
            s["power"] = read(hw + "/power1_average") or read(hw + "/power1_input")
            s["temp"] = read(hw + "/temp1_input")
            s["temp_mem"] = read(hw + "/temp3_input")
        samples.append(s)
        time.sleep(0.02)


res = ix.new_result()
ix.timing_enable(True)
th = threading.Thread(target=sampler)
th.start()
prev, rows = {}, []
t_start = time.perf_counter()
for i in range(22):
    ix.run(reads, res, fulgor_amd.FULL_INTERSECTION, 0.0, 0, 10000000)
    res.expand()
    t = ix.timing()
    cur = {k: v[0] for k, v in t.items()}
    rows.append({k: cur[k] - prev.get(k, 0.0) for k in ("k1_lookup", "k2_intersect", "k2b_expand")})
    prev = cur
t_end = time.perf_counter()
stop = True
th.join()
rows = rows[2:]
print("passes 3..22: " + ", ".join("%s median %.3f min %.3f max %.3f ms" % (k, statistics.median(r[k] for r in rows), min(r[k] for r in rows), max(r[k] for r in rows))
                                    for k in ("k1_lookup", "k2_intersect", "k2b_expand")))
busy = [s for s in samples if t_start <= s["t"] <= t_end]


def num(x):
    try:
        return float("".join(ch for ch in x if ch.isdigit() or ch == "."))
    except ValueError:
        return float("nan")


if busy:
    for key in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "power", "temp", "temp_mem"):
        vals = [num(s.get(key, "")) for s in busy if s.get(key)]
        if vals:
            print("  %-12s during the passes: median %.0f  min %.0f  max %.0f  (%d samples)" % (key, statistics.median(vals), min(vals), max(vals), len(vals)))
print(sh("rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v '^=\\|^$'").strip())
del res, reads
ix.close()
for exe in ("hbm_rates", "store_patterns"):
    p = os.path.join(ROOT, "profiles", "micro", exe)
    if os.path.exists(p):
        print("-- %s" % exe)
        print(sh(p).strip())
