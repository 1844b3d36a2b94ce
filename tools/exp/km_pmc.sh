#!/bin/bash
# HBM bytes of the k-means kernels from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only; KiB units;
# FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, as tools/pmc_traffic.py does for the training step).
# usage (repo root, through gpurun): tools/exp/km_pmc.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/kmpmc_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/kmpmc_$C -o b -- \
    python $R/bench.py --workload kmeans --kmeans-data mixture --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
done
cd $R
python - <<PY > gpurun_out/${TAG}_km_pmc.txt
import csv, glob, collections, json, re
ITER = 6          # --steps 4 --warmup 2
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("gpurun_out/kmpmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            agg[k][0] += 1
            agg[k][1] += float(r.get("Counter_Value", 0) or 0)
    out[c] = agg
names = [k for k in out["FETCH_SIZE"] if "km_" in k or "kmeans" in k or "csplit" in k]
short = lambda k: re.sub(r"^void |\(anonymous namespace\)::", "", k).split("(")[0][:44]
print("# HBM MB per launch (FETCH_SIZE x 2 x 1024 B, WRITE_SIZE x 1024 B), 'bench.py --workload kmeans --kmeans-data mixture --steps 4 --warmup 2'")
print("%-46s %8s %12s %12s" % ("kernel", "launches", "fetch MB", "write MB"))
per_iter = 0.0
for k in sorted(names, key=lambda k: -out["FETCH_SIZE"][k][1]):
    n, f = out["FETCH_SIZE"][k]
    w = out["WRITE_SIZE"].get(k, [1, 0.0])
    fm, wm = f / n * 2 * 1024 / 1e6, w[1] / max(w[0], 1) * 1024 / 1e6
    print("%-46s %8d %12.1f %12.1f" % (short(k), n, fm, wm))
    if n >= ITER:
        per_iter += (fm + wm) * (n // ITER)
print("# one Lloyd iteration (every kernel launched once or twice per iteration): %.1f MB" % per_iter)
json.dump({"kmeans_iteration_hbm_bytes": per_iter * 1e6,
           "note": "tools/exp/km_pmc.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of 'python bench.py --workload "
                   "kmeans --kmeans-data mixture --no-cpu-baseline --steps 4 --warmup 2'; KiB units, FETCH_SIZE doubled per the gfx950 correction "
                   "of MI355X_MICROARCH.md; the kernels of one iteration summed"}, open("gpurun_out/${TAG}_km_pmc_traffic.json", "w"), indent=1)
PY
rm -rf gpurun_out/kmpmc_FETCH_SIZE gpurun_out/kmpmc_WRITE_SIZE
cat gpurun_out/${TAG}_km_pmc.txt
