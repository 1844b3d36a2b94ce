#!/bin/bash
# Per-kernel time per step of the serial training step for the kernels whose name contains one of the given substrings.
# usage (repo root, through gpurun): tools/exp/kernel_times.sh <tag> <substring> [...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-rXX}; shift
$R/tools/exp/census.sh $TAG > /dev/null
head -9 $R/gpurun_out/${TAG}_step_census_serial.txt
python - "$R/gpurun_out/${TAG}_train_serial/bench_kernel_stats.csv" "$@" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in sys.argv[2:]):
        print("%-80s calls/step %5.1f  ms/step %7.3f" % (r["Name"][:80], int(r["Calls"]) / 5, float(r["TotalDurationNs"]) / 1e6 / 5))
PY
