#!/bin/bash
# Per-kernel durations of the default bench on the GPU box (rocprofv3 kernel trace):
#   gpurun -- bash profiles/ab_trace.sh <tag> [bench args...]
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$tag
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
f=$(ls "$OUT"/trace/*kernel_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && { echo "trace failed"; tail -5 "$OUT/trace.err"; exit 1; }
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.1f} us  {float(r["Percentage"]):5.1f} %')
P
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
