#!/bin/bash
# Idle time of the overlapped training step (tools/step_idle.py): gpurun_out/<tag>_step_idle.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_idle -o bench -- \
  python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/${TAG}_idle.log 2>&1
cd $R/tools && python step_idle.py $R/gpurun_out/${TAG}_idle 12 ${2:-0} ${3:-0} > $R/gpurun_out/${TAG}_step_idle.txt
rm -rf $R/gpurun_out/${TAG}_idle
head -20 $R/gpurun_out/${TAG}_step_idle.txt
