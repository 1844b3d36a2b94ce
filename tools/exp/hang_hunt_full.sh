#!/bin/bash
# The driver's exact bench command (training line + extra workloads + CPU baseline) and the inference workload alone, repeated; a run that
# exceeds its limit gets SIGABRT and leaves the python stacks (faulthandler).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for i in $(seq 1 ${1:-4}); do
  PYTHONFAULTHANDLER=1 timeout -s ABRT 400 python bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/full_$i.txt 2>/tmp/fullerr_$i.txt
  echo "full $i rc=$? $(tail -1 /tmp/full_$i.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), [round(e['value'],4) for e in d.get('extra_workloads',[]) if isinstance(e,dict)])" 2>&1 | tail -1)"
done
for i in $(seq 1 ${2:-12}); do
  PYTHONFAULTHANDLER=1 timeout -s ABRT 90 python bench.py --workload infer --no-cpu-baseline > /tmp/inf_$i.txt 2>/tmp/inferr_$i.txt
  RC=$?
  echo "infer $i rc=$RC $(tail -1 /tmp/inf_$i.txt | cut -c1-0)$(tail -1 /tmp/inf_$i.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1))" 2>&1 | tail -1)"
  if [ $RC -ne 0 ]; then grep -v "dist-packages" /tmp/inferr_$i.txt | tail -25; fi
done
