#!/bin/bash
# A/B timing of engine builds on ONE box (boxes differ by a few percent): usage  gpu_ab_probe.sh libA.so libB.so ...
# prints the per-kernel averages of the six tile kernels and the steady step time, two interleaved rounds.
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib (round $round)"
    CHGNET_HIP_LIB=$lib timeout 150 python tools/gpu_scale_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady" | awk '{printf "%s %s | ", $1, $(NF-1)} END {print ""}'
  done
done
