#!/bin/bash
# HBM bytes of the EM kernel at the stress shape (256 images), set-up alone (num_iter 1) and the whole run (50), tiled / row-by-row
# pair pass: FETCH_SIZE and WRITE_SIZE in their own passes (kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/pmc_em_traffic
rm -rf $R; mkdir -p $R
for it in 1 50; do for mode in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/p_${it}_${mode}_$c -o c -- python3 scripts/em_traffic_probe.py $it $mode > $R/p_${it}_${mode}_$c.log 2>&1
  echo "num_iter $it mode $mode $c:" $(python3 scripts/rocpd_pmc.py $R/p_${it}_${mode}_$c/c_results.db em_batch 2>&1 | tail -1 | cut -c1-200)
done; done; done
find $R -name '*.db' -delete
