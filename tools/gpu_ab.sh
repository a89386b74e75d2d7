#!/bin/bash
# A/B on ONE box: libraries chgnet_amd/lib/libchgnet_hip<v>.so for v in $AB_VARIANTS (default: "_prev" and the product build), twice each
mkdir -p gpurun_out/ab
for rep in 1 2; do
for v in ${AB_VARIANTS:-_prev ""}; do
  [ "$v" = "cur" ] && v=""
  echo "=== lib$v (rep $rep)" | tee -a gpurun_out/ab/log.txt
  CHGNET_HIP_LIB=$PWD/chgnet_amd/lib/libchgnet_hip$v.so timeout 240 python tools/gpu_scale_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady" | tee -a gpurun_out/ab/log.txt
done
done
