#!/bin/bash
# Runs on the GPU box (through gpurun) from the repository root:
#   bash profiles/run_profile.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of the default bench (per-kernel durations)
# 2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) for HBM traffic per launch
# Summaries are written to gpurun_out/<tag>/ and condensed by profiles/summarize.py.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
# (the profiled process may fault inside the tool's exit handlers after every output file has been
# written: judge the pass by its output, not by its exit status)
ls "$OUT"/trace/*kernel_stats.csv > /dev/null 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.err"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/fetch_bench.json" 2> "$OUT/fetch.err"
find "$OUT/pmc_fetch" -name "*counter_collection.csv" | grep -q . || { echo "fetch pass failed"; tail -5 "$OUT/fetch.err"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/write_bench.json" 2> "$OUT/write.err"
find "$OUT/pmc_write" -name "*counter_collection.csv" | grep -q . || { echo "write pass failed"; tail -5 "$OUT/write.err"; exit 1; }
python3 "$ROOT/profiles/summarize.py" "$OUT" > "$OUT/summary.md"
# keep the merged-back payload small: drop the raw per-dispatch tables
find "$OUT" -name "*counter_collection.csv" -size +20M -delete
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
cat "$OUT/summary.md"
