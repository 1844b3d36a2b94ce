#!/bin/bash
mkdir -p gpurun_out
S=tests/native/selftest
LD_LIBRARY_PATH=tests/native/oldlib timeout 600 $S bench2 0x2000000 0x2003000 0x2004000 > gpurun_out/r3e9_nofence.log 2>&1
timeout 600 $S bench2 0x2000000 0x2003000 0x2004000 > gpurun_out/r3e9_fence.log 2>&1
LD_LIBRARY_PATH=tests/native/oldlib timeout 600 $S bench2 0x2000000 0x2003000 0x2004000 > gpurun_out/r3e9_nofence2.log 2>&1
timeout 600 $S bench2 0x2000000 0x2003000 0x2004000 > gpurun_out/r3e9_fence2.log 2>&1
