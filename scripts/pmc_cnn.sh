#!/bin/bash
# PMC counters for the CNN kernels alone (own passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/pmc_cnn
rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $R/p1 -o c -- python3 scripts/time_cnn.py 102 > $R/p1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES -d $R/p2 -o c -- python3 scripts/time_cnn.py 102 > $R/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d $R/p3 -o c -- python3 scripts/time_cnn.py 102 > $R/p3.log 2>&1
for p in p1 p2 p3; do python3 scripts/rocpd_pmc.py $(ls $R/$p/*/*.db | head -1) conv_gemm > $R/$p.txt 2>&1; done
cat $R/p1.txt $R/p2.txt $R/p3.txt | cut -c1-600
