#!/bin/bash
# Team-mode angle adjoints: parity (every small batch through them) and MD-size per-kernel times.
O=gpurun_out/r06/team; mkdir -p $O
CHGNET_TEAM_MIN_ANGLES=0 timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_team0.log 2>&1; echo "pytest (team forced) exit $?"; tail -3 $O/pytest_team0.log
for sc in 2,2,2 4,2,2; do
  CHGNET_HIP_GRAPHS=0 timeout 300 python tools/gpu_md_kernel_probe.py 50 $sc > $O/md_probe_$sc.log 2>&1; echo "md probe $sc exit $?"; head -8 $O/md_probe_$sc.log
  CHGNET_TEAM_MIN_ANGLES=-1 CHGNET_HIP_GRAPHS=0 timeout 300 python tools/gpu_md_kernel_probe.py 50 $sc > $O/md_probe_off_$sc.log 2>&1; head -3 $O/md_probe_off_$sc.log
done
timeout 300 python tools/gpu_md_anatomy.py 300 > $O/md_anatomy.log 2>&1; head -12 $O/md_anatomy.log
timeout 300 python tools/gpu_md_anatomy.py 300 4,2,2 > $O/md_anatomy_512.log 2>&1; head -3 $O/md_anatomy_512.log
