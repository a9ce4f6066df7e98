#!/bin/bash
# Everything the numbers of a round are taken from, in one call on the GPU box (from the repository
# root): bash profiles/final_measure.sh <tag>
#   GPU tests, smoke, the default bench (with the CPU baseline), the other shapes and the second
#   camera model, the kernel trace with the two traffic passes, the SQ counter pass.
# A step that fails ends the script.
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$OUT/tests.log" 2>&1 || { tail -20 "$OUT/tests.log"; exit 1; }
tail -2 "$OUT/tests.log"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1 || { tail -20 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
timeout -k 10 600 python bench.py > "$OUT/bench_1gpu.json" 2> "$OUT/bench_1gpu.err" || { tail -5 "$OUT/bench_1gpu.err"; exit 1; }
for shape in fountain ladybug final; do
    timeout -k 10 600 python bench.py --shape $shape --no-cpu-baseline > "$OUT/bench_$shape.json" 2> "$OUT/bench_$shape.err" || { tail -5 "$OUT/bench_$shape.err"; exit 1; }
done
timeout -k 10 600 python bench.py --camera-model bal --no-cpu-baseline > "$OUT/bench_bal.json" 2> "$OUT/bench_bal.err" || { tail -5 "$OUT/bench_bal.err"; exit 1; }
bash profiles/run_profile.sh "$TAG/prof" > /dev/null || exit 1
bash profiles/pmc_sq.sh "$TAG/sq" > "$OUT/sq_counters.md" || exit 1
echo "measured: $OUT"
