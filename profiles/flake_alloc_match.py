"""reads the stderr of a faulted run with FULGOR_TRACE_ALLOC=1: the device buffers that end at or just below the faulting address"""
import re, sys
text = open(sys.argv[1], errors="replace").read()
m = re.search(r"on address (0x[0-9a-f]+)", text)
if not m:
    print("no fault"); sys.exit(0)
addr = int(m.group(1), 16)
print("fault at %#x" % addr)
live = {}
for a, b, want, asked, obj in re.findall(r"\[alloc\] (0x[0-9a-f]+) \.\. (0x[0-9a-f]+) \((\d+) bytes for (\d+) asked\) buffer object (0x[0-9a-f]+)", text):
    live[obj] = (int(a, 16), int(b, 16), int(want), int(asked))
near = sorted((addr - b, a, b, want, asked, obj) for obj, (a, b, want, asked) in live.items() if -(8 << 20) < addr - b < (8 << 20))
for d, a, b, want, asked, obj in near[:12]:
    print("  buffer %s: %#x .. %#x (%d bytes, %d asked) ends %d bytes %s the fault" % (obj, a, b, want, asked, abs(d), "below" if d >= 0 else "above"))
