#!/bin/bash
# same-box A/B of library variants: AB_VARIANTS="_prev cur" (default), each twice; the GPU parity suite on the product build first
R=$PWD; O=$R/gpurun_out/ab2; mkdir -p $O; rm -f $O/ab.log
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | head -8
for rep in 1 2; do
for v in ${AB_VARIANTS:-_prev cur}; do
  [ "$v" = "cur" ] && v=""
  echo "=== lib$v (rep $rep)" | tee -a $O/ab.log
  CHGNET_HIP_LIB=$R/chgnet_amd/lib/libchgnet_hip$v.so timeout 200 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|embed|gemm|steady" | tee -a $O/ab.log
done
done
