#!/bin/bash
mkdir -p gpurun_out/t2
for v in "" _t2_NO_DUMP _t2_NO_ROWATOM _t2_NO_ROWBWD _t2_ALL; do
  echo "=== lib$v" | tee -a gpurun_out/t2/variants.txt
  CHGNET_HIP_LIB=$PWD/chgnet_amd/lib/libchgnet_hip$v.so timeout 240 python tools/gpu_train_probe.py 1024 2>&1 | grep -E "t2_atom|t2_bond|t2_angle|profiled backward" | grep -v first | tee -a gpurun_out/t2/variants.txt
done
