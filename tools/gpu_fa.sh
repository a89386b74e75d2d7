#!/bin/bash
# per-atom AngleUpdate forward (kernels_angle_fa.h) against the row-order kernel, same library, same box; GPU suite first
R=$PWD; O=$R/gpurun_out/fa; mkdir -p $O; rm -f $O/ab.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |Error" | head -12 | tee $O/pytest.log
for rep in 1 2; do
for v in 0 1; do
  echo "=== CHGNET_PER_ATOM_FWD=$v (rep $rep)" | tee -a $O/ab.log
  CHGNET_PER_ATOM_FWD=$v timeout 200 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady" | tee -a $O/ab.log
done
done
