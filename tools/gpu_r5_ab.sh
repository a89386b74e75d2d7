#!/bin/bash
# same-box A/B of variant libraries: bash tools/gpu_r5_ab.sh exp prio1 prio3 ...   (chgnet_amd/lib/libchgnet_hip_<name>.so; "product" = the product library)
R=$PWD
for rep in 1 2; do
for v in "$@"; do
  lib=$R/chgnet_amd/lib/libchgnet_hip_$v.so; [ "$v" = product ] && lib=$R/chgnet_amd/lib/libchgnet_hip.so
  echo "== $v (pass $rep)"; CHGNET_HIP_LIB=$lib python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady" | awk '{printf "%s %s | ", $1, $NF=="ms"?$(NF-1):$0} END{print ""}'
done; done
