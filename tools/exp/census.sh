#!/bin/bash
# Serial kernel census of one training step (the dispatches between two optimizer launches): gpurun_out/<tag>_step_census_serial.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_train_serial -o bench -- \
  python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra --serial > $R/gpurun_out/${TAG}_train_serial.log 2>&1
python $R/tools/step_census.py $R/gpurun_out/${TAG}_train_serial > $R/gpurun_out/${TAG}_step_census_serial.txt
find $R/gpurun_out -name "*kernel_trace.csv" -path "*${TAG}_*" -delete
head -12 $R/gpurun_out/${TAG}_step_census_serial.txt
