#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/r3pmcKA -o p -- python $R/bench.py --workload kmeans --kmeans-data randn --no-cpu-baseline --steps 4 --warmup 1 > $R/gpurun_out/r3pmcKA.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r3pmcKB -o p -- python $R/bench.py --workload kmeans --kmeans-data randn --no-cpu-baseline --steps 4 --warmup 1 > $R/gpurun_out/r3pmcKB.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/r3pmcKA | grep "screen\|kernel,counter" 
python $R/tools/pmc_summary.py $R/gpurun_out/r3pmcKB | grep "screen\|kernel,counter"
find $R/gpurun_out/r3pmcKA $R/gpurun_out/r3pmcKB -name "*.csv" -delete
