#!/bin/bash
# Same-box A/B helper: build chgnet_amd/lib/libchgnet_hip_<name>.so from a COPY of csrc (a git ref, or the work tree) after applying
# an optional sed script.  usage: tools/build_alt_lib.sh <name> <ref|WORK> [sed-script]
set -e
name=$1; ref=$2; script=$3
R=$(cd "$(dirname "$0")/.." && pwd)
S=/tmp/alt_$name; rm -rf $S; mkdir -p $S/obj
if [ "$ref" = "WORK" ]; then cp $R/chgnet_amd/csrc/* $S/; else (cd $R && git archive $ref chgnet_amd/csrc | tar -x -C $S --strip-components=2); fi
[ -n "$script" ] && sed -i -E "$script" $S/*.h $S/*.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -Wno-unused-function -Wno-unused-value -Wno-unused-result -I$R/include -I$S"
for u in engine engine_predict engine_train engine_graph comm; do /opt/rocm/bin/hipcc $FLAGS -c $S/$u.hip -o $S/obj/$u.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $S/obj/*.o -o $R/chgnet_amd/lib/libchgnet_hip_$name.so
echo built $R/chgnet_amd/lib/libchgnet_hip_$name.so
