#!/bin/bash
# A/B of engine builds on one box: parity subset first (each variant must be CORRECT), then interleaved timing.
# usage: tools/gpu_ab_run.sh out_dir libA.so libB.so ...
O=$1; shift; mkdir -p $O
for lib in "$@"; do
  tag=$(basename $lib .so)
  CHGNET_HIP_LIB=$PWD/$lib timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or stage_buffers or mixed or ragged or full_size" > $O/parity_$tag.log 2>&1
  echo "parity $tag rc=$? $(tail -1 $O/parity_$tag.log)"
done
libs=""; for lib in "$@"; do libs="$libs $PWD/$lib"; done
bash tools/gpu_ab_probe.sh $libs 2>&1 | tee $O/ab.log
