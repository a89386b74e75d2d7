#!/bin/bash
mkdir -p gpurun_out/exp2
for v in _q _noatom_q _noatom_nogather_q; do
  echo "=== lib$v" | tee -a gpurun_out/exp2/log.txt
  CHGNET_HIP_LIB=$PWD/chgnet_amd/lib/libchgnet_hip$v.so timeout 240 python tools/gpu_order_probe.py 1024 2>&1 | grep -v Warning | tee -a gpurun_out/exp2/log.txt
done
