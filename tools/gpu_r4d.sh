#!/bin/bash
R=$PWD; O=$R/gpurun_out/r4d; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log; grep -E "^E |Error" $O/pytest_gpu.log | head -20
timeout 300 python bench.py --steps 5 --warmup 2 --no-configs --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err; tail -c 1500 $O/bench_quick.json
