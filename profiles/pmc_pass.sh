#!/bin/bash
# One counter pass over one bench solve, summarised per kernel:
#   bash profiles/pmc_pass.sh <tag> COUNTER [COUNTER ...]
# (stay within the per-block limits of one pass: 8 SQ, 4 TCP, 2 TA, 2 TD, 4 TCC, 2 GRBM)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc" -o pass -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/pass.err"
find "$OUT/pmc" -name "*counter_collection.csv" | grep -q . || { echo "pass failed"; tail -5 "$OUT/pass.err"; exit 1; }
python3 - "$OUT" "$@" <<'PY' | tee "$OUT/summary.md"
import csv, glob, os, sys
from collections import defaultdict
out, names = sys.argv[1], sys.argv[2:]
path = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0].replace("sfmba::", "").replace("void ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("per launch | kernel | n | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get(names[0], [0]))):
    n = max(len(x) for x in v.values())
    print(f"| {k} | {n} | " + " | ".join(f"{sum(v.get(c, [0])) / max(n, 1):.4g}" for c in names) + " |")
PY
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
