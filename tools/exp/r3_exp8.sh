#!/bin/bash
mkdir -p gpurun_out
S=tests/native/selftest
# halo kernel: default / no setprio / no sched_barrier / neither / never-nt stores
timeout 900 $S bench2 0x1000000 0x21000000 0x41000000 0x61000000 0x1200000 > gpurun_out/r3e8_bench2.log 2>&1
