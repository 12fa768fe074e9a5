#!/bin/bash
# counters of the LRN+pool kernels (own passes, kernel-trace only; every pass under its own timeout)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/pmc_lrn
rm -rf $R; mkdir -p $R
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/p1 -o c -- python3 scripts/time_cnn.py 102 > $R/p1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/p2 -o c -- python3 scripts/time_cnn.py 102 > $R/p2.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY -d $R/p3 -o c -- python3 scripts/time_cnn.py 102 > $R/p3.log 2>&1
for p in p1 p2 p3; do python3 scripts/rocpd_pmc.py $R/$p/c_results.db ${1:-lrn} | cut -c1-600; done
