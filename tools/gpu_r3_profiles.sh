#!/bin/bash
# Round-3 rocprofv3 evidence (GPU box): kernel stats + HBM counters of the headline step, kernel stats of one training step
# and of the MD loop, VALU counters of the embedding kernels.  Counters and traces in separate runs.
TAG=${1:-r03}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O/prof
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ktrace -- $B > $O/rocprof_ktrace.log 2>&1; echo "ktrace exit $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof -o pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $O/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof -o pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $O/rocprof_write.log 2>&1; echo "pmc write exit $?"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof -o pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs > $O/rocprof_sq.log 2>&1; echo "pmc sq exit $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o train -- python $R/tools/gpu_train_probe.py 1024 > $O/rocprof_train.log 2>&1; echo "train ktrace exit $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof -o train_fetch -- python $R/tools/gpu_train_probe.py 1024 > $O/rocprof_train_fetch.log 2>&1; echo "train pmc fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof -o train_write -- python $R/tools/gpu_train_probe.py 1024 > $O/rocprof_train_write.log 2>&1; echo "train pmc write exit $?"
# MD loop: eager launches (rocprofv3's kernel trace segfaults on hipGraph replays of this size)
CHGNET_HIP_GRAPHS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o md -- python $R/tools/gpu_md_probe.py 200 > $O/rocprof_md.log 2>&1; echo "md ktrace exit $?"
ls $O/prof | head -30
