#!/bin/bash
# Per kernel of the shipped gfx950 code object: registers, spills, scratch, LDS, and the share of lane moves (v_readlane / v_writelane:
# what SGPR spills turn into) among the static VALU instructions. With --classes <kernel substring>: the VALU instructions of the
# matching kernels split into the two issue classes measured in profiles/r1/valu_rates_r1q.txt (about 2.5 cycles: add, sub, and, or,
# xor, lshr, mov, not, bfe/bfi-free moves; about 4.2-4.6: everything VOP3-only, shift-left, min/max, compares, multiplies, alignbit,
# perm, readlane/writelane, cndmask).
#   profiles/isa_report.sh [lib.so] [--classes k1_lookup]
set -e
LIB=fulgor_amd/libfulgor_gpu.so
CLASSES=""
while [ $# -gt 0 ]; do
  case "$1" in
    --classes) CLASSES="$2"; shift 2;;
    *) LIB="$1"; shift;;
  esac
done
BIN=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
trap 'rm -rf $TMP' EXIT
$BIN/llvm-objcopy --dump-section .hip_fatbin=$TMP/fatbin.bin $LIB
TGT=$($BIN/clang-offload-bundler --list --type=o --input=$TMP/fatbin.bin | grep gfx950 | head -1)
$BIN/clang-offload-bundler --unbundle --type=o --input=$TMP/fatbin.bin --targets=$TGT --output=$TMP/code.co
$BIN/llvm-readelf --notes $TMP/code.co > $TMP/notes.txt
$BIN/llvm-objdump -d $TMP/code.co > $TMP/dis.txt
python3 - "$TMP/notes.txt" "$TMP/dis.txt" "$CLASSES" <<'PY'
import re, subprocess, sys
notes, dis, classes = open(sys.argv[1]).read(), open(sys.argv[2]).read(), sys.argv[3]
def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n
meta = {}
for blk in notes.split("- .agpr_count:")[1:]:
    f = dict(re.findall(r"\.(\w+):\s+('?[\w.$@]+'?)", "agpr_count: " + blk.split("\n", 1)[0] + "\n" + blk))
    name = f.get("name", "").strip("'")
    meta[name] = f
# static instruction counts per kernel symbol
counts = {}
cur = None
CHEAP = re.compile(r"^v_(add|sub|subrev|and|or|xor|lshrrev|mov|not|xnor)_(u32|i32|b32|co_u32|nc_u32)\b|^v_(add|sub|subrev)_co_u32\b")
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
    if m:
        cur = m.group(1)
        counts[cur] = {"valu": 0, "lane": 0, "salu": 0, "cheap": 0, "vmem": 0, "lds": 0}
        continue
    if cur is None:
        continue
    m = re.match(r"^\s+(\w+)", line)
    if not m:
        continue
    op = m.group(1)
    c = counts[cur]
    if op.startswith("v_"):
        c["valu"] += 1
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            c["lane"] += 1
        base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
        if CHEAP.match(base) and "_e64" not in op and "_dpp" not in op:
            c["cheap"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c["vmem"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
print("%-64s %5s %5s %6s %6s %7s %6s %6s %6s %9s" % ("kernel", "vgpr", "sgpr", "sspill", "vspill", "scratch", "lds", "VALU", "SALU", "lane moves"))
for name in sorted(meta, key=lambda n: demangle(n)):
    f = meta[name]
    sym = name[:-3] if name.endswith(".kd") else name
    c = counts.get(sym, {"valu": 0, "lane": 0, "salu": 0})
    d = demangle(sym)
    d = re.sub(r"\(.*", "", d)[:64]
    print("%-64s %5s %5s %6s %6s %7s %6s %6d %6d %5d %3.0f%%" % (d, f.get("vgpr_count"), f.get("sgpr_count"), f.get("sgpr_spill_count"), f.get("vgpr_spill_count"),
          f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size"), c["valu"], c["salu"], c["lane"], 100.0 * c["lane"] / max(1, c["valu"])))
if classes:
    print()
    for sym, c in counts.items():
        d = demangle(sym)
        if classes in d and c["valu"]:
            print("%s: %d static VALU = %d in the 2.5-cycle class (plain add/sub/and/or/xor/lshr/mov/not in VOP1/VOP2 form) + %d in the 4.2-4.6-cycle class; %d SALU, %d vector memory, %d LDS"
                  % (re.sub(r"\(.*", "", d), c["valu"], c["cheap"], c["valu"] - c["cheap"], c["salu"], c["vmem"], c["lds"]))
PY
