#!/bin/bash
# A/B: norm passes with / without the nt hint on tensors > 160 MB (second library built with NT_BYTES = inf)
cp u2seg_amd/csrc/libu2seg_hip.so /tmp/lib_default.so
for lib in /tmp/lib_default.so tests/native/oldlib/libu2seg_hip_nont.so /tmp/lib_default.so tests/native/oldlib/libu2seg_hip_nont.so; do
cp $lib u2seg_amd/csrc/libu2seg_hip.so
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), round(d['ms_per_step'],2))"
done
cp /tmp/lib_default.so u2seg_amd/csrc/libu2seg_hip.so
