#!/bin/bash
# occupancy probe: forward tile kernels with ONE wave per SIMD (4-wave workgroups, one per CU) against the product's two
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
export CHGNET_HIP_LIB=$R/chgnet_amd/lib/libchgnet_hip_exp.so CHGNET_PER_ATOM_FWD=0
echo "== 2 waves per SIMD (product shape)"; CHGNET_EXP_NW=0 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "atomconv_fwd|bondconv_fwd|angleupd_fwd|steady"
echo "== 1 wave per SIMD"; CHGNET_EXP_NW=4 python tools/gpu_kernel_probe.py 1024 2>&1 | grep -E "atomconv_fwd|bondconv_fwd|angleupd_fwd|steady"
