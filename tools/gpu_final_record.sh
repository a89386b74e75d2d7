#!/bin/bash
# bench line and the headline step's rocprofv3 kernel stats / HBM counters from ONE box (box-to-box spread is 3-5 %)
TAG=${1:-r03}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O/prof
timeout 420 python bench.py > $R/gpurun_out/bench_line.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ktrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $O/rocprof_ktrace.log 2>&1; echo "ktrace exit $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof -o pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $O/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof -o pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $O/rocprof_write.log 2>&1; echo "pmc write exit $?"
tail -c 300 $R/gpurun_out/bench_line.json
