#!/bin/bash
# One environment setting of the library against problems (GPU box):
#   bash profiles/study_env.sh VAR "v1 v2" "prob1 prob2" [engine options]
for p in $3; do
  for v in $2; do
    env $1=$v timeout -k 10 120 python -u profiles/study_trajectory.py $p $4 2>&1 | grep "final" | grep -v Function | sed "s/^/$1=$v  /"
  done
done
