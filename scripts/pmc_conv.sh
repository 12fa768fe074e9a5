#!/bin/bash
# PMC counters for the conv kernels (own pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc1 -o conv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES -d gpurun_out/pmc2 -o conv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2.log 2>&1
ls -la gpurun_out/pmc1 gpurun_out/pmc2
