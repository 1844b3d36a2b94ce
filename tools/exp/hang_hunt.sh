#!/bin/bash
# Repeats the driver's bench command; a run that exceeds 100 s gets SIGABRT and leaves the python stacks of all threads (faulthandler).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
N=${1:-8}
for i in $(seq 1 $N); do
  S=$(date +%s.%N)
  PYTHONFAULTHANDLER=1 timeout -s ABRT ${HANG_T:-40} python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > /tmp/out_$i.txt 2>/tmp/err_$i.txt
  RC=$?
  E=$(date +%s.%N)
  echo "run $i rc=$RC $(tail -1 /tmp/out_$i.txt | cut -c1-120)"
  if [ $RC -ne 0 ]; then grep -v "^  File.*site-packages\|^  File.*dist-packages" /tmp/err_$i.txt | tail -60; fi
done
