#!/bin/bash
# One environment switch of the library over several bench configurations (GPU box):
#   bash profiles/ab_env_args.sh VAR "v1 v2" "--shape final" "--shape final --scale 0.2" ...
# prints ms per solve, LM and PCG iterations and the final cost for every (configuration, value).
VAR=$1; VALS=$2; shift 2
for args in "$@"; do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$args', '$VAR=$v', round(d['ms_per_step'],3), 'ms', d['lm_iterations_per_solve'], 'LM', d['pcg_iterations_per_solve'], 'PCG', repr(d['final_cost']))" || exit 1
  done
done
