#!/bin/bash
# Full GPU evidence run (GPU box): tests, smoke, bench (+CPU baseline), rocprofv3 kernel stats and PMC passes.
R=$PWD; O=$R/gpurun_out; mkdir -p $O/prof
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 280 python bench.py --steps 10 --warmup 3 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-600
CHGNET_BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_dist1.log 2>&1; echo "force-dist exit $?"; tail -1 $O/bench_dist1.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ktrace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_ktrace.log 2>&1; echo "ktrace exit $?"
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof -o pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof -o pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/rocprof_write.log 2>&1; echo "pmc write exit $?"
ls -la $O/prof | head -30
