#!/bin/bash
# SQ-level PMC passes for the hot kernels (GPU box).  Counters only, no tracing domains.
R=$PWD; O=$R/gpurun_out/prof2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O -o sq1 -- python $R/tools/gpu_scale_probe.py 512 > $O/sq1.log 2>&1; echo "sq1 $?"
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O -o sq2 -- python $R/tools/gpu_scale_probe.py 512 > $O/sq2.log 2>&1; echo "sq2 $?"
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES --output-format csv -d $O -o sq3 -- python $R/tools/gpu_scale_probe.py 512 > $O/sq3.log 2>&1; echo "sq3 $?"
ls $O
