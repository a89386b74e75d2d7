#!/bin/bash
# cross-compiles here (no GPU needed); run the binary on the GPU box: chgnet_amd/lib/split_lab
set -e
cd "$(dirname "$0")/.."
mkdir -p chgnet_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=fast -DCHG_SPLIT_LO_SEPARATE=1 -Ichgnet_amd/csrc -Iinclude tools/split_lab.hip -o chgnet_amd/lib/split_lab
