#!/bin/bash
# wait / instruction counters of the raster kernels (own passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/pmc_raster
rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES -d $R/p1 -o c -- python3 scripts/time_raster.py 2 > $R/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $R/p2 -o c -- python3 scripts/time_raster.py 2 > $R/p2.log 2>&1
for k in simplify outline coverage blend; do for p in p1 p2; do python3 scripts/rocpd_pmc.py $R/$p/c_results.db $k 2>&1 | cut -c1-900; done; done
tail -2 $R/p1.log | cut -c1-300
