#!/bin/bash
R=$PWD; O=$R/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_v020.py tests/test_gpu_train.py -x -q -m gpu > $O/test_v020_train.log 2>&1; echo "exit $?"; tail -15 $O/test_v020_train.log
