#!/bin/bash
# fused dE/d h_bond update in the AtomConv adjoint against table + gemm_GQ, same library, same box; GPU suite first
R=$PWD; O=$R/gpurun_out/fuse; mkdir -p $O; rm -f $O/ab.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |Error" | head -12 | tee $O/pytest.log
for rep in 1 2; do
for v in 0 1; do
  echo "=== CHGNET_FUSE_GQ=$v (rep $rep)" | tee -a $O/ab.log
  CHGNET_FUSE_GQ=$v timeout 200 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "atomconv_bwd|gemm_GQ|gemm_GP|steady" | tee -a $O/ab.log
done
done
for v in 0 1; do echo "=== MD CHGNET_FUSE_GQ=$v"; CHGNET_FUSE_GQ=$v MD_PROBE_REPS=200 timeout 200 python tools/gpu_md_breakdown.py ef 2>&1 | grep -E "replay predict  |launches,"; done
