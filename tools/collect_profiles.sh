#!/bin/bash
# Round profiles on the GPU box (run from the repo root through gpurun): kernel-trace statistics of the three bench.py
# workloads and the two HBM-traffic PMC passes of the training step, all under gpurun_out/<tag>_*.
#   usage: tools/collect_profiles.sh <tag>        e.g. tools/collect_profiles.sh r02
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_train -o bench -- \
  python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > $R/gpurun_out/${TAG}_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_train_serial -o bench -- \
  python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra --serial > $R/gpurun_out/${TAG}_train_serial.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o bench -- \
  python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --serial > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o bench -- \
  python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --serial > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kmeans -o bench -- \
  python $R/bench.py --workload kmeans --no-cpu-baseline > $R/gpurun_out/${TAG}_kmeans.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_infer -o bench -- \
  python $R/bench.py --workload infer --no-cpu-baseline > $R/gpurun_out/${TAG}_infer.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_fetch > $R/gpurun_out/${TAG}_pmc_fetch_size_summary.csv
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_write > $R/gpurun_out/${TAG}_pmc_write_size_summary.csv
python $R/tools/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_fetch $R/gpurun_out/${TAG}_pmc_write $R/gpurun_out/${TAG}_pmc_traffic.json \
  "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra --serial\` (tools/collect_profiles.sh ${TAG}); KiB units; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md; WRITE_SIZE uncalibrated" > /dev/null
find $R/gpurun_out/${TAG}_pmc_fetch $R/gpurun_out/${TAG}_pmc_write -name "*.csv" -delete
python $R/tools/step_census.py $R/gpurun_out/${TAG}_train > $R/gpurun_out/${TAG}_step_census.txt
python $R/tools/step_census.py $R/gpurun_out/${TAG}_train_serial > $R/gpurun_out/${TAG}_step_census_serial.txt
# keep the merge small: the per-dispatch traces are not needed, only the statistics and the counter rows
find $R/gpurun_out -name "*kernel_trace.csv" -path "*${TAG}_*" ! -path "*pmc*" -delete
ls -la $R/gpurun_out/${TAG}_*/ | head -40
for f in train train_serial pmc_fetch pmc_write kmeans infer; do tail -1 $R/gpurun_out/${TAG}_$f.log | cut -c1-200; done
