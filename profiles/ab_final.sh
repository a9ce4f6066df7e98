#!/bin/bash
# Variants of the solver at the Final shape, one trajectory and four timed solves each:
#   bash profiles/ab_final.sh <out tag> VAR=value[,VAR=value...] ...
out=gpurun_out/$1; mkdir -p $out; shift
for v in "$@"; do
    echo "== $v"
    env ${v//,/ } timeout -k 10 250 python -u profiles/study_final_shape.py 3 2>&1 | grep -v amdgpu.ids || exit 1
done > $out/final_ab.txt 2>&1
