#!/bin/bash
# second-order training sweep: parity tests + timing of the fused against the unfused pipeline (one box)
mkdir -p gpurun_out/t2
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/t2/tests.txt
timeout 300 python tools/gpu_train_probe.py 1024 2>&1 | grep -v "first-order" | tee gpurun_out/t2/probe_fused.txt
if [ -n "$T2_AB" ]; then CHGNET_T2_UNFUSED=1 timeout 300 python tools/gpu_train_probe.py 1024 2>&1 | grep "second-order" | tee gpurun_out/t2/probe_unfused.txt; fi
