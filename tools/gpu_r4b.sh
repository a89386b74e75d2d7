#!/bin/bash
R=$PWD; O=$R/gpurun_out/r4b; mkdir -p $O/prof
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for v in "" ; do
  echo "=== lib$v" | tee -a $O/ab.log
  CHGNET_HIP_LIB=$R/chgnet_amd/lib/libchgnet_hip$v.so timeout 200 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady" | tee -a $O/ab.log
done
echo "=== cur, row-order forward" | tee -a $O/ab.log
CHGNET_PER_ATOM_FWD=0 timeout 200 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady" | tee -a $O/ab.log
