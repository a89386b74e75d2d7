#!/bin/bash
# L2 (TCC) counters of the headline step: hit rate and fabric requests per kernel (two --pmc passes, no trace domains)
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O/prof
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_ATOMIC_sum --output-format csv -d $O/prof -o l2a -- $B > $O/rocprof_l2a.log 2>&1; echo "l2a exit $?"
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/prof -o l2b -- $B > $O/rocprof_l2b.log 2>&1; echo "l2b exit $?"
ls $O/prof | grep l2
cd /tmp
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/prof -o l2c -- $B > $O/rocprof_l2c.log 2>&1; echo "l2c exit $?"
