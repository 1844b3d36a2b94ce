#!/bin/bash
# The round's closing measurements on one box: profiles (tools/collect_profiles.sh), the driver's bench command, the per-layer table.
# usage (repo root, through gpurun): tools/exp/round_end.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
cd $R
tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
cd $R
timeout -s KILL 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.log 2>&1
tail -1 gpurun_out/${TAG}_bench_default.log > gpurun_out/${TAG}_bench_default.json
timeout -s KILL 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra --serial --per-layer > gpurun_out/${TAG}_per_layer.txt 2>&1
tail -3 gpurun_out/${TAG}_collect.log | cut -c1-300
cut -c1-600 gpurun_out/${TAG}_bench_default.json
