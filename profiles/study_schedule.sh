#!/bin/bash
# PCG-tolerance schedules against problems (GPU box): bash profiles/study_schedule.sh "prob1 prob2" "sched1 sched2" [engine options]
for p in $1; do
  for s in $2; do
    SFMBA_RTOL_SCHEDULE=$s timeout -k 10 120 python -u profiles/study_trajectory.py $p $3 2>&1 | grep "final" | sed "s/^/$s  /"
  done
done
