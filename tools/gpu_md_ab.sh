#!/bin/bash
# MD-size A/B on ONE box: libchgnet_hip_prev.so vs the product build -- replay / eager step of the 256-atom cell, then the headline kernels
mkdir -p gpurun_out/ab
: > gpurun_out/ab/md_log.txt
for rep in 1 2; do
for v in _prev ""; do
  echo "=== lib$v (rep $rep)" | tee -a gpurun_out/ab/md_log.txt
  CHGNET_HIP_LIB=$PWD/chgnet_amd/lib/libchgnet_hip$v.so MD_PROBE_REPS=200 timeout 240 python tools/gpu_md_breakdown.py efs 2>&1 | grep -E "replay predict  |build \+ eager|bondconv|angleupd|atomconv" | tee -a gpurun_out/ab/md_log.txt
done
done
for v in _prev ""; do
  echo "=== lib$v headline kernels" | tee -a gpurun_out/ab/md_log.txt
  CHGNET_HIP_LIB=$PWD/chgnet_amd/lib/libchgnet_hip$v.so timeout 240 python tools/gpu_scale_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady" | tee -a gpurun_out/ab/md_log.txt
done
