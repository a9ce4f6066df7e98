"""GPU idle time between the kernels of the solves in a rocprofv3 kernel trace (study script):
python profiles/gap_analysis.py <dir with *kernel_trace.csv>.  Kernels closer than 1 ms to their
predecessor count as one burst; per burst: span, busy time, idle time, number of idle gaps > 5 us."""
import csv
import glob
import os
import sys

path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
bursts, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - cur[-1][1] > 1_000_000:
        bursts.append(cur)
        cur = []
    cur.append(e)
bursts.append(cur)
print("| burst | kernels | span us | busy us | idle us | gaps > 5 us | largest gaps us (before kernel) |")
print("|---|---|---|---|---|---|---|")
for i, b in enumerate(bursts):
    if len(b) < 20:
        continue
    span = b[-1][1] - b[0][0]
    busy = sum(e[1] - e[0] for e in b)
    gaps = [(b[j][0] - b[j - 1][1], b[j][2].split("(")[0].replace("sfmba::", "").replace("void ", "")[:28]) for j in range(1, len(b))]
    big = sorted((g for g in gaps if g[0] > 5000), reverse=True)
    print(f"| {i} | {len(b)} | {span / 1e3:.1f} | {busy / 1e3:.1f} | {(span - busy) / 1e3:.1f} | {len(big)} | "
          + ", ".join(f"{g[0] / 1e3:.1f} ({g[1]})" for g in big[:6]) + " |")
