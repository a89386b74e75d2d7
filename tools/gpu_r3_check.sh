#!/bin/bash
# parity subset + per-kernel timing of the current build (one box)
mkdir -p gpurun_out/check
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/check/parity.txt
timeout 300 python tools/gpu_scale_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|gemm_|embed|steady|end to end" | tee gpurun_out/check/probe.txt
