#!/bin/bash
# Kernel timeline of a RESIDENT MD-size prediction (256 / 512 atoms): per-kernel durations and the gaps between consecutive dispatches.
#   tools/gpu_md_trace.sh [tag]  ->  gpurun_out/<tag>/md_trace_{256,512}.txt (+ raw csv)
TAG=${1:-r06/trace}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for sc in 2,2,2 4,2,2; do
  for G in 1 0; do
    CHGNET_HIP_GRAPHS=$G timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_${sc}_g$G -o t -- python $R/tools/gpu_md_replay_probe.py 200 $sc > $O/run_${sc}_g$G.log 2>&1
    echo "trace $sc graphs=$G exit $?"; tail -1 $O/run_${sc}_g$G.log
    python $R/tools/md_trace_summary.py $O/prof_${sc}_g$G > $O/md_trace_${sc}_g$G.txt 2>&1
    rm -rf $O/prof_${sc}_g$G
  done
done
cat $O/md_trace_2,2,2_g1.txt
