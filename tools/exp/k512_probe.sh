#!/bin/bash
# what holds the K >= 512 1x1 layers at 2.6-3.0 TB/s?  Isolated launches of the layer through the C ABI
# (tests/native/selftest bench2), timing per kernel variant and counter passes (SQ, TCC, TCP; one --pmc pass each, --kernel-trace
# only).  usage (from the repo root, through gpurun): tools/exp/k512_probe.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/${TAG}_k512
mkdir -p $OUT
cd $R
hipcc -O2 --offload-arch=gfx950 tests/native/selftest.cpp -Iinclude -Lu2seg_amd/csrc -lu2seg_hip -Wl,-rpath,$R/u2seg_amd/csrc -o tests/native/selftest 2> $OUT/build.log || { tail -5 $OUT/build.log; exit 1; }
ST=$R/tests/native/selftest
# 1. timing: automatic dispatch, stream-K forbidden, 256x128 tiles with two groups per CU (cfg 4), ring 5 (cfg 2)
for L in "res4 1x1 1024->256 plain" "res4 1x1 256->1024 plain" "res5 1x1 2048->512" "res5 1x1 512->2048" "lat3 1x1 512->256" "p4 3x3 256->256 50x84" "res2 1x1 64->256 plain"; do
  U2_BENCH_LAYERS="$L" $ST bench2 0 0x10000000 0x4000 0x2000 | grep LAYER
done > $OUT/timing.txt 2>&1
cat $OUT/timing.txt
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC|TA|TD|SQ|GRBM)_[A-Z0-9_]+" | sort -u > $OUT/counters_available.txt
wc -l $OUT/counters_available.txt
cd /tmp && export TMPDIR=/tmp
pass() {  # name, layer, counters...
  local name=$1 layer=$2; shift 2
  U2_BENCH_LAYERS="$layer" timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- $ST bench2 0 > $OUT/pmc_$name.log 2>&1
  python $R/tools/pmc_summary.py $OUT/pmc_$name | grep -v "^\"\(void \)\?\(fill\|__amd\)" | head -14 > $OUT/pmc_$name.csv
  rm -rf $OUT/pmc_$name
}
for LN in "a:res4 1x1 1024->256 plain" "b:res5 1x1 2048->512" "c:p4 3x3 256->256 50x84" "d:res2 1x1 64->256 plain"; do
  T=${LN%%:*}; L=${LN#*:}
  pass ${T}_sq "$L" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
  pass ${T}_tcc1 "$L" TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  pass ${T}_tcc2 "$L" TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_READ_sum
  pass ${T}_tcc3 "$L" TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum
  pass ${T}_tcp1 "$L" TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
  pass ${T}_tcp2 "$L" TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
  pass ${T}_fetch "$L" FETCH_SIZE
done
tail -n +1 $OUT/pmc_*.csv | cut -c1-200
