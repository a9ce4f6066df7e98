"""Time line of the last solve in a rocprofv3 kernel trace (csv): per kernel the launches, the time
inside the kernel and the idle time in front of it (start minus the end of the kernel before).
    python profiles/timeline.py <kernel_trace.csv>
"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void sfmba::", "")))
rows.sort()
# a solve ends with the gradient-only linearisation (k_lin_cm<..., true>) and the scalars after it:
# the last solve lies between the last two of those
ends = [i for i, r in enumerate(rows) if "k_lin_cm<" in r[2] and ", true>" in r[2]]
start = 0
if len(ends) >= 2:
    start = ends[-2] + 1
    while "k_scal_finalize" in rows[start][2] or "fillBuffer" in rows[start][2]:
        start += 1
    rows = rows[:ends[-1] + 2]
sel = rows[start:]
busy = defaultdict(float); idle = defaultdict(float); n = defaultdict(int)
for i, (s, e, k) in enumerate(sel):
    busy[k] += (e - s) / 1e3
    n[k] += 1
    if i:
        idle[k] += max(0, s - sel[i - 1][1]) / 1e3
total = (sel[-1][1] - sel[0][0]) / 1e3
print(f"last solve: {len(sel)} launches, {total:.1f} us from first start to last end, "
      f"{sum(busy.values()):.1f} us inside kernels, {sum(idle.values()):.1f} us idle between them")
print(f"{'kernel':60s} {'n':>5s} {'busy us':>10s} {'avg':>8s} {'idle before':>12s} {'avg':>7s}")
for k in sorted(busy, key=lambda k: -(busy[k] + idle[k])):
    print(f"{k[:60]:60s} {n[k]:5d} {busy[k]:10.1f} {busy[k] / n[k]:8.2f} {idle[k]:12.1f} {idle[k] / n[k]:7.2f}")
