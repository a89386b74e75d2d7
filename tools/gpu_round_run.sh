#!/bin/bash
# Full GPU evidence run (GPU box): tests, smoke, bench (+CPU baseline, all configs), rocprofv3 kernel stats and PMC passes.
# usage: tools/gpu_round_run.sh [tag]     -> gpurun_out/<tag>/..., summarise afterwards with profiles/summarize.py <tag>
TAG=${1:-r02}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O/prof
timeout 600 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 500 python bench.py --steps 10 --warmup 3 > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ktrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs > $O/rocprof_ktrace.log 2>&1; echo "ktrace exit $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof -o pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $O/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof -o pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $O/rocprof_write.log 2>&1; echo "pmc write exit $?"
ls $O/prof | head -20
