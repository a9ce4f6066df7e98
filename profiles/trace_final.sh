#!/bin/bash
# Kernel trace of solves at the Final shape, condensed into a time line of the last solve:
#   bash profiles/trace_final.sh <tag>      (on the GPU box, from the repository root)
set -u
TAG=${1:-final}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (SFMBA_STUDY_SHAPE in the environment selects another shape)
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/profiles/study_final_shape.py" 0 > "$OUT/trace.out" 2> "$OUT/trace.err"
f=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "trace failed"; tail -5 "$OUT/trace.err"; exit 1; }
python3 "$ROOT/profiles/timeline.py" "$f" > "$OUT/timeline.txt"
cp "$f" "$OUT/kernel_trace.csv"; rm -rf "$OUT/trace"
tail -40 "$OUT/timeline.txt"
