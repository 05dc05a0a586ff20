"""Timeline of a rocprofv3 --kernel-trace run: per kernel, the mean duration and the mean idle gap between the end of the
previous kernel on the device and its own start -- where the wall-clock of a launch-bound loop (the SCP's oracle round
trips) goes beyond the kernels themselves.   usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [skip_first_n]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[skip:]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = None
for s, e, name in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    short = short.split("<")[0] + ("<" + short.split("<", 1)[1] if "<" in short else "")
    dur[short] += e - s
    cnt[short] += 1
    if prev_end is not None:
        g = s - prev_end
        if g < 2_000_000:                      # (ignore pauses longer than 2 ms: set-up between phases)
            gap[short] += max(g, 0)
    prev_end = max(e, prev_end or e)
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, busy {sum(dur.values()) / 1e6:.3f} ms")
print(f"{'kernel':58s} {'calls':>6s} {'mean us':>9s} {'gap before us':>14s} {'total ms':>9s} {'gaps ms':>8s}")
for k in sorted(dur, key=lambda k: -dur[k]):
    print(f"{k[:58]:58s} {cnt[k]:6d} {dur[k] / cnt[k] / 1e3:9.2f} {gap[k] / cnt[k] / 1e3:14.2f} {dur[k] / 1e6:9.3f} {gap[k] / 1e6:8.3f}")
n_dump = int(os.environ.get("TRACE_DUMP", "0"))
if n_dump:   # the raw sequence at the END of the trace (the last SCP iterations): start (us), duration, idle gap before, kernel
    tail = rows[-n_dump:]
    t0 = tail[0][0]
    prev = None
    print(f"--- last {n_dump} kernels: start us | duration us | gap before us | kernel")
    for s, e, name in tail:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {((s - prev) / 1e3 if prev else 0):8.1f}  {short}")
        prev = e
