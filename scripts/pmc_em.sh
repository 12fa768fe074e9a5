#!/bin/bash
# instruction-cache and wait counters of the EM kernel (own passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/pmc_em
rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $R/p1 -o c -- python3 scripts/em_breakdown.py > $R/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_WAIT_ANY SQ_INSTS_SALU -d $R/p2 -o c -- python3 scripts/em_breakdown.py > $R/p2.log 2>&1
for p in p1 p2; do python3 scripts/rocpd_pmc.py $R/$p/c_results.db em_batch > $R/$p.txt 2>&1; done
cat $R/p1.txt $R/p2.txt | cut -c1-700; tail -3 $R/p1.log | cut -c1-300
