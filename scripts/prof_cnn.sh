#!/bin/bash
# kernel trace of the CNN alone (scripts/time_cnn.py, 2 warm-up + 8 timed passes at B = 102) -> per-kernel averages (dev tool)
#   VPK_ALGORITHM=2 bash scripts/prof_cnn.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-cnn}
R=gpurun_out/prof_$T
rm -rf $R; mkdir -p $R
timeout 300 rocprofv3 --kernel-trace --stats -d $R/trace -o t -- python3 scripts/time_cnn.py --passes 10 102 > $R/run.log 2>&1
python3 scripts/rocpd_stats.py $(find $R/trace -name '*.db' | head -1) $R/kernel_stats.csv --skip-passes 2 --passes 10 > $R/top.txt
cut -c1-150 $R/top.txt | head -24
grep "B=102" $R/run.log | cut -c1-300
find $R -name '*.db' -delete
