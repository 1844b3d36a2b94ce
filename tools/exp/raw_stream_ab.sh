#!/bin/bash
# The binding's stream argument from torch._C._cuda_getCurrentRawStream (default) or from a torch.cuda.Stream object per launch
# (U2_HIP_STREAM_OBJECT=1): driver's bench command, alternating.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for P in 0 1 0 1 0 1 0 1; do
  U2_HIP_STREAM_OBJECT=$P timeout -s KILL 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('U2_HIP_STREAM_OBJECT=$P', round(d['value'],1), 'img/s', d['per_step']['device_ms_overlapped_steps'])"
done
