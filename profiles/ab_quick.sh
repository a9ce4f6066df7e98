#!/bin/bash
# Quick A/B on the GPU box: the parity tests that exercise the kernels, then one bench line.
#   gpurun -- bash profiles/ab_quick.sh <tag> [bench args...]
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_irregular.py tests/test_gpu_bal.py -m gpu -x -q \
    --timeout 120 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -2 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit $?
python - <<P
import json
d = json.load(open('gpurun_out/${tag}_bench.json'))
print(round(d['ms_per_step'], 4), d['lm_iterations_per_solve'], d['pcg_iterations_per_solve'], d['final_cost'])
for k, v in d['kernels'].items():
    if v['launches']:
        print('   ', k, v['launches'], round(v['total_ms'], 3), round(v['avg_us'], 1))
P
