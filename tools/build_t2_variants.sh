#!/bin/bash
# timing-only variants of the fused second-order kernels (wrong results): which phase bounds k2_atom<true>
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-value -Wno-unused-result -Iinclude -Ichgnet_amd/csrc"
for v in NO_DUMP NO_ROWATOM NO_ROWBWD; do
  /opt/rocm/bin/hipcc $FLAGS -DCHG_EXPERIMENTS -DCHG_EXP_T2_$v chgnet_amd/csrc/engine.hip chgnet_amd/csrc/comm.hip -o chgnet_amd/lib/libchgnet_hip_t2_$v.so
done
/opt/rocm/bin/hipcc $FLAGS -DCHG_EXPERIMENTS -DCHG_EXP_T2_NO_DUMP -DCHG_EXP_T2_NO_ROWATOM -DCHG_EXP_T2_NO_ROWBWD chgnet_amd/csrc/engine.hip chgnet_amd/csrc/comm.hip -o chgnet_amd/lib/libchgnet_hip_t2_ALL.so
