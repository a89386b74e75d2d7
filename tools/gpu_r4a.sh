#!/bin/bash
# round 4, first GPU run: parity suite, then same-box A/B of the libraries and the L2 / HBM counters of the new kernels
R=$PWD; O=$R/gpurun_out/r4a; mkdir -p $O/prof
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
for v in _r03 _nointer ""; do
  echo "=== lib$v" | tee -a $O/ab.log
  CHGNET_HIP_LIB=$R/chgnet_amd/lib/libchgnet_hip$v.so timeout 200 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady|gemm_G|embed" | tee -a $O/ab.log
done
echo "=== cur, row-order forward" | tee -a $O/ab.log
CHGNET_PER_ATOM_FWD=0 timeout 200 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady" | tee -a $O/ab.log
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof -o pmc_fetch -- $B > $O/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof -o pmc_write -- $B > $O/rocprof_write.log 2>&1; echo "pmc write exit $?"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_ATOMIC_sum --output-format csv -d $O/prof -o l2a -- $B > $O/rocprof_l2a.log 2>&1; echo "l2a exit $?"
ls $O/prof | head
