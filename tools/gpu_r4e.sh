#!/bin/bash
R=$PWD; O=$R/gpurun_out/r4e; mkdir -p $O
run() { echo "=== $*" | tee -a $O/ab.log; env "$@" timeout 200 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "conv_|angleupd_|steady" | tee -a $O/ab.log; }
run CHGNET_TILE_INTERLEAVE=27
run CHGNET_TILE_INTERLEAVE=31
run CHGNET_TILE_INTERLEAVE=0
run CHGNET_TILE_INTERLEAVE=27 CHGNET_GRID_MULT=1
run CHGNET_TILE_INTERLEAVE=27 CHGNET_GRID_MULT=4
run CHGNET_TILE_INTERLEAVE=27
