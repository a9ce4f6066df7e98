#!/bin/bash
# Solve time of the other bench shapes with two builds of the library (A/B on the GPU box):
#   gpurun -- bash profiles/ab_shapes.sh <other-library.so>
for lib in "$1" ""; do
  for args in "--shape fountain" "--shape ladybug" "--shape final" "--camera-model bal" "--shape venice --seed 1" "--shape final --scale 0.2"; do
    SFMBA_LIBRARY=$lib timeout -k 10 200 python bench.py --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${lib:-current}'.split('/')[-1], '$args', round(d['ms_per_step'],3), d['lm_iterations_per_solve'], d['pcg_iterations_per_solve'], round(d['kernels']['linearize']['avg_us'],1))" || exit 1
  done
done
