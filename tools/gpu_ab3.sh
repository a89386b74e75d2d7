#!/bin/bash
# same-box A/B of library variants without the test suite: AB_VARIANTS="_prev cur" (default), AB_REPS (2), AB_ENV="VAR=1 ..." per run
R=$PWD; O=$R/gpurun_out/ab3; mkdir -p $O; rm -f $O/ab.log
for rep in $(seq 1 ${AB_REPS:-2}); do
for v in ${AB_VARIANTS:-_prev cur}; do
  [ "$v" = "cur" ] && v=""
  echo "=== lib$v (rep $rep)" | tee -a $O/ab.log
  env $AB_ENV CHGNET_HIP_LIB=$R/chgnet_amd/lib/libchgnet_hip$v.so timeout 200 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "${AB_GREP:-conv_|angleupd_|steady}" | tee -a $O/ab.log
done
done
