"""Busy time, idle gaps and launch count of the kernel stream from a rocprofv3 --kernel-trace csv.
usage: trace_gaps.py <dir with *_kernel_trace.csv> [first_fraction]   (looks at the last 40 % of the trace: the timed steps)"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
f = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
n = len(ev)
ev = ev[int(n * 0.6):]
span = ev[-1][1] - ev[0][0]
busy = 0
gaps = []
cur_end = ev[0][0]
for s, e, k in ev:
    if s > cur_end:
        gaps.append(s - cur_end)
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
print("kernels %d  span %.2f ms  busy %.2f ms (%.1f %%)  gaps: n=%d total %.2f ms, median %.2f us, >20us: %d (%.2f ms)" % (
    len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, len(gaps), sum(gaps) / 1e6,
    sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))
per = collections.defaultdict(lambda: [0, 0])
for s, e, k in ev:
    per[k][0] += 1
    per[k][1] += e - s
short = sum(c for k, (c, t) in per.items() if t / c < 10000)
print("launches with mean < 10 us: %d of %d, their time %.2f ms" % (short, len(ev), sum(t for k, (c, t) in per.items() if t / c < 10000) / 1e6))
