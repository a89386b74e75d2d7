#!/bin/bash
# Round-6 record from ONE box (boxes differ by 2-4 %): the bench line, then rocprofv3 kernel stats, HBM counters (separate --pmc passes),
# L2 hit counters and SQ counters of the SAME command; kernel stats of one training step and of the MD loop.
#   tools/gpu_round6_profiles.sh [tag]  ->  gpurun_out/<tag>/...;  afterwards: python profiles/summarize.py <tag>
TAG=${1:-r06}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O/prof
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 400 $O/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ktrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $O/rocprof_ktrace.log 2>&1; echo "ktrace exit $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof -o pmc_fetch -- $B > $O/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof -o pmc_write -- $B > $O/rocprof_write.log 2>&1; echo "pmc write exit $?"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_ATOMIC_sum --output-format csv -d $O/prof -o l2a -- $B > $O/rocprof_l2a.log 2>&1; echo "l2a exit $?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof -o pmc_sq -- $B > $O/rocprof_sq.log 2>&1; echo "pmc sq exit $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train -- python $R/tools/gpu_train_probe.py 1024 > $O/rocprof_train.log 2>&1; echo "train ktrace exit $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o md -- python $R/tools/gpu_md_probe.py 400 > $O/rocprof_md.log 2>&1; echo "md ktrace exit $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o md512 -- python $R/tools/gpu_md_anatomy.py 300 4,2,2 > $O/rocprof_md512.log 2>&1; echo "md512 ktrace exit $?"
# the MD loop without the profiler, new path and the launch sequence of the large batches, alternating (same box)
for i in 1 2 3; do
  timeout 300 python $R/tools/gpu_md_anatomy.py 500 2>&1 | grep "steps/s" | sed "s/^/new path: /"
  CHGNET_TINY_FUSE=0 CHGNET_TEAM_MIN_ANGLES=-1 CHGNET_BLK_MAX_ANGLES=0 timeout 300 python $R/tools/gpu_md_anatomy.py 500 2>&1 | grep "steps/s" | sed "s/^/r05 launch sequence: /"
done > $O/md_ab.txt 2>&1; cat $O/md_ab.txt
ls $O/prof | head -30
