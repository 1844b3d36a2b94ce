cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kmprof -o bench -- python $R/bench.py --workload kmeans --kmeans-data mixture --no-cpu-baseline --steps 20 --warmup 5 > $R/gpurun_out/kmprof.log 2>&1
f=$(ls $R/gpurun_out/kmprof/*kernel_stats.csv | head -1)
head -14 $f | cut -c1-160
