#!/bin/bash
# timing-only: every sigmoid's exp / rcp replaced by full-rate arithmetic (WRONG results) -- the share of the quarter-rate transcendentals
R=$PWD
for v in exp notrans; do
  echo "== $v"; CHGNET_HIP_LIB=$R/chgnet_amd/lib/libchgnet_hip_$v.so python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady"
done
