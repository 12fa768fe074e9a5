#!/bin/bash
# kernel trace of a short bench run -> timeline of its last steps (dev tool)   bash scripts/prof_timeline.sh "--em-wgs 160 --em-slice-ms 2.5"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/prof_tl
rm -rf $R; mkdir -p $R
timeout 300 rocprofv3 --kernel-trace -d $R/trace -o t -- python3 bench.py --no-extra --no-cpu-baseline --no-alt --steps 12 --warmup 4 $1 > $R/run.log 2>&1
python3 scripts/rocpd_timeline.py $(find $R/trace -name '*.db' | head -1) 16 > $R/timeline.txt
head -90 $R/timeline.txt
find $R -name '*.db' -delete
