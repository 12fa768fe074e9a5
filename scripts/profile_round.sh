#!/bin/bash
# rocprofv3 passes for profiles/: kernel trace (+stats db) and HBM traffic counters, separate passes;
# summaries land in gpurun_out/prof/summary (copy them to profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/prof
rm -rf $R; mkdir -p $R/summary
rocprofv3 --kernel-trace --stats -d $R/yud_trace -o t -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/yud_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/stress_trace -o t -- python3 bench.py --workload stress --steps 3 --warmup 1 --no-cpu-baseline > $R/stress_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/cnn_trace -o t -- python3 scripts/time_cnn.py 102 > $R/cnn_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/yud_fetch -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/yud_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/yud_write -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/yud_write.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/stress_fetch -o t -- python3 bench.py --workload stress --steps 2 --warmup 1 --no-cpu-baseline > $R/stress_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/stress_write -o t -- python3 bench.py --workload stress --steps 2 --warmup 1 --no-cpu-baseline > $R/stress_write.log 2>&1
N=${ROUND:-r01}
for w in yud stress cnn; do
  python3 scripts/rocpd_stats.py $(find $R/${w}_trace -name '*.db' | head -1) $R/summary/${N}_${w}_kernel_stats.csv > $R/summary/${N}_${w}_top.txt
done
python3 scripts/make_traffic_json.py $R $R/summary/${N}_pmc_traffic.json > /dev/null
grep -h '"metric"' $R/yud_trace.log $R/stress_trace.log | cut -c1-300
cat $R/summary/${N}_yud_top.txt | head -8
find $R -name '*.db' -delete
du -sh $R
