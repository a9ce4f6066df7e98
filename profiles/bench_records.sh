#!/bin/bash
# The bench lines kept under profiles/ for a round, in one go (on the GPU box, from the repository root):
#   bash profiles/bench_records.sh <tag>     ->  gpurun_out/<tag>/bench_*.json
set -u
TAG=${1:-bench}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { timeout -k 10 400 python bench.py "$@" 2>> $OUT/bench.err | tail -1; }
run > $OUT/bench_1gpu.json || exit 1
run --steps 20 > $OUT/bench_20steps.json || exit 1
{ run --shape fountain --no-cpu-baseline && run --shape ladybug --no-cpu-baseline && run --shape temple --no-cpu-baseline &&
  run --shape final --steps 3 --warmup 1 --no-cpu-baseline && run --camera-model bal --no-cpu-baseline &&
  run --camera-model bal --shape ladybug --no-cpu-baseline; } > $OUT/bench_other_shapes.json || exit 1
{ run --shape fountain --through-class --no-cpu-baseline && run --shape temple --through-class --no-cpu-baseline; } > $OUT/bench_through_class.json || exit 1
run --shape final --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_final_shape.json || exit 1
python - $OUT <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line)
            print(f.split("/")[-1], d["config"]["workload"][:44], "ms", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 4),
                  "traffic", d["roofline"].get("traffic"), "class", d.get("optimize_class_seconds"))
PY
