#!/bin/bash
# PMC on the fpn_output2 layer: halo kernel vs tile kernel (selftest `one <variant>`)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 16777216 8192; do
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/r3pmcA_$v -o p -- $R/tests/native/selftest one $v > $R/gpurun_out/r3pmcA_$v.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/r3pmcB_$v -o p -- $R/tests/native/selftest one $v > $R/gpurun_out/r3pmcB_$v.log 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/r3pmcA_$v > $R/gpurun_out/r3pmcA_$v.txt
  python $R/tools/pmc_summary.py $R/gpurun_out/r3pmcB_$v > $R/gpurun_out/r3pmcB_$v.txt
  find $R/gpurun_out/r3pmcA_$v $R/gpurun_out/r3pmcB_$v -name "*.csv" -delete
  cat $R/gpurun_out/r3pmcA_$v.txt $R/gpurun_out/r3pmcB_$v.txt
done
