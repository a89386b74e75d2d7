#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -x -q -m gpu > $O/test_wide.log 2>&1; echo "exit $?"; tail -25 $O/test_wide.log
python tools/gpu_kernel_probe.py 1024 2>&1 | tail -3
