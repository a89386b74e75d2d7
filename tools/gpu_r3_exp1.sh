#!/bin/bash
# round-3 experiment 1: angle order x atomics/gather bounds, and phase timing (one box)
mkdir -p gpurun_out/exp1
for v in "" _noatom _nogather _noatom_nogather; do
  echo "=== lib$v" | tee -a gpurun_out/exp1/log.txt
  CHGNET_HIP_LIB=$PWD/chgnet_amd/lib/libchgnet_hip$v.so timeout 240 python tools/gpu_order_probe.py 1024 2>&1 | grep -v Warning | tee -a gpurun_out/exp1/log.txt
done
echo "=== phase" | tee -a gpurun_out/exp1/log.txt
CHGNET_HIP_LIB=$PWD/chgnet_amd/lib/libchgnet_hip_phase.so timeout 240 python tools/gpu_order_probe.py 1024 phase 2>&1 | grep -v Warning | tee -a gpurun_out/exp1/log.txt
