#!/bin/bash
# Round-5 record from ONE box (boxes differ by 2-4 %): the bench line, then rocprofv3 kernel stats, HBM counters (separate --pmc passes),
# L2 hit counters and SQ counters of the SAME command; kernel stats of one training step and of the MD loop.
#   tools/gpu_round5_profiles.sh [tag]  ->  gpurun_out/<tag>/...;  afterwards: python profiles/summarize.py <tag>
TAG=${1:-r05}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O/prof
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 400 $O/bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ktrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $O/rocprof_ktrace.log 2>&1; echo "ktrace exit $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof -o pmc_fetch -- $B > $O/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof -o pmc_write -- $B > $O/rocprof_write.log 2>&1; echo "pmc write exit $?"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_ATOMIC_sum --output-format csv -d $O/prof -o l2a -- $B > $O/rocprof_l2a.log 2>&1; echo "l2a exit $?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof -o pmc_sq -- $B > $O/rocprof_sq.log 2>&1; echo "pmc sq exit $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train -- python $R/tools/gpu_train_probe.py 1024 > $O/rocprof_train.log 2>&1; echo "train ktrace exit $?"
CHGNET_HIP_GRAPHS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o md -- python $R/tools/gpu_md_probe.py 200 > $O/rocprof_md.log 2>&1; echo "md ktrace exit $?"
ls $O/prof | head -30
