#!/bin/bash
# the GPU suite N times; prints the summary line and whatever failed
N=${1:-4}
for i in $(seq 1 $N); do
timeout 400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "^FAILED|^E  |passed|failed" | cut -c1-300 | tail -12
done
