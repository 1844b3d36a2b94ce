#!/bin/bash
# Per-launch durations of the ROIAlign kernels in one serial training step (kernel trace).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/roi_trace -o t -- \
  python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra --serial > /dev/null 2>&1
python - $R/gpurun_out/roi_trace <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "roi_align" in r["Kernel_Name"]]
for r in sel[-16:]:
    print("%-60s grid %-22s %8.1f us" % (r["Kernel_Name"][:60], "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $R/gpurun_out/roi_trace
