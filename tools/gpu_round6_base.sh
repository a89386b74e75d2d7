#!/bin/bash
# Round-6 baseline on one box: GPU tests, the bench line, MD-size per-kernel times (plain adjoints and per-atom adjoints).
O=gpurun_out/r06/base; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench exit $?"; tail -c 600 $O/bench_line.json; echo
for sc in 2,2,2 4,2,2; do
  CHGNET_HIP_GRAPHS=0 timeout 300 python tools/gpu_md_kernel_probe.py 50 $sc > $O/md_probe_$sc.log 2>&1; echo "md probe $sc exit $?"
  CHGNET_HIP_GRAPHS=0 CHGNET_WIN_MIN_ATOMS_PER_WAVE=0 timeout 300 python tools/gpu_md_kernel_probe.py 50 $sc > $O/md_probe_win_$sc.log 2>&1; echo "md probe win $sc exit $?"
done
timeout 300 python tools/gpu_md_anatomy.py 300 > $O/md_anatomy.log 2>&1
cat $O/md_probe_2,2,2.log
