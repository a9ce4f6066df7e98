#!/bin/bash
# SQ counter pass over one bench solve: bash profiles/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -o sq -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/sq_bench.json" 2> "$OUT/sq.err"
# (the profiled process may fault in the tool's exit handlers after the output is complete)
find "$OUT/pmc_sq" -name "*counter_collection.csv" | grep -q . || { echo "sq pass failed"; tail -5 "$OUT/sq.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
path = glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("sfmba::", "").replace("void ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS"]
print("| kernel | n | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    n = len(v.get("SQ_WAVE_CYCLES", []))
    print(f"| {k} | {n} | " + " | ".join(f"{sum(v.get(c, [0])) / max(n, 1):.3g}" for c in names) + " |")
PY
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
