#!/bin/bash
# SQ counters of one workload (two --pmc passes, --kernel-trace only), summarised per kernel by tools/pmc_summary.py.
#   usage: tools/pmc_kernel.sh <tag> <command...>      (run from the repo root through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_a -o p -- "$@" > $R/gpurun_out/${TAG}_pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM \
  --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_b -o p -- "$@" > $R/gpurun_out/${TAG}_pmc_b.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_a > $R/gpurun_out/${TAG}_pmc_a_summary.csv
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_pmc_b > $R/gpurun_out/${TAG}_pmc_b_summary.csv
find $R/gpurun_out/${TAG}_pmc_a $R/gpurun_out/${TAG}_pmc_b -name "*.csv" -size +2M -delete
head -12 $R/gpurun_out/${TAG}_pmc_a_summary.csv; head -12 $R/gpurun_out/${TAG}_pmc_b_summary.csv
