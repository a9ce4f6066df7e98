#!/bin/bash
# FETCH_SIZE per launch of the large kernels, one bench solve: bash profiles/pmc_fetch_quick.sh <tag>
TAG=${1:-fq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/fetch_bench.json" 2> "$OUT/fetch.err"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
path = glob.glob(os.path.join(sys.argv[1], "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(list)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0].replace("sfmba::", "").replace("void ", "")].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"{k[:50]:50s} n {len(v):3d} mean KiB {sum(v)/len(v):10.1f} min {min(v):10.1f} max {max(v):10.1f}")
PY
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
