#!/bin/bash
# rocprofv3 passes for profiles/ (one MI355X through gpurun): kernel traces (+ per-kernel stats), HBM traffic
# counters and MFMA-busy counters, every --pmc set in its own pass with --kernel-trace only.
# Summaries land in gpurun_out/prof/summary (copy them to profiles/).   ROUND=r03 bash scripts/profile_round.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/prof
N=${ROUND:-r06}
rm -rf $R; mkdir -p $R/summary
BENCH="python3 bench.py --workload yud --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/yud_trace -o t -- $BENCH > $R/yud_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/stress_trace -o t -- python3 bench.py --workload stress --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $R/stress_trace.log 2>&1
# CNN alone: 2 warm-up + 12 timed forward passes at B = 102 (the summary drops the warm-up dispatches)
timeout 300 rocprofv3 --kernel-trace --stats -d $R/cnn_trace -o t -- python3 scripts/time_cnn.py --passes 14 102 > $R/cnn_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/yud_fetch -o t -- python3 bench.py --workload yud --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $R/yud_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/yud_write -o t -- python3 bench.py --workload yud --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $R/yud_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/stress_fetch -o t -- python3 bench.py --workload stress --steps 2 --warmup 1 --no-cpu-baseline > $R/stress_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/stress_write -o t -- python3 bench.py --workload stress --steps 2 --warmup 1 --no-cpu-baseline > $R/stress_write.log 2>&1
# round 5's first default (conv2 / fc6 on exact bf16 triples, conv3..5 Winograd on the f32 cores)
export VPK_ALGORITHM=2
timeout 300 rocprofv3 --kernel-trace --stats -d $R/cnn_triples_trace -o t -- python3 scripts/time_cnn.py --passes 14 102 > $R/cnn_triples_trace.log 2>&1
unset VPK_ALGORITHM
# CNN alone with the f32 direct kernels everywhere (vpk_cnn_set_fusion(1), vpk_cnn_set_algorithm(0): rounds 1-3's default) and with
# round 4's defaults (f32 direct conv1, Winograd conv2..5 on the f32 matrix cores)
export VPK_ALGORITHM=0 VPK_FUSION=1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/cnn_direct_trace -o t -- python3 scripts/time_cnn.py --passes 14 102 > $R/cnn_direct_trace.log 2>&1
export VPK_ALGORITHM=1 VPK_FUSION=1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/cnn_wino_trace -o t -- python3 scripts/time_cnn.py --passes 14 102 > $R/cnn_wino_trace.log 2>&1
unset VPK_ALGORITHM VPK_FUSION
# CNN alone with conv2..5 on the bf16 matrix cores (vpk_cnn_set_precision(1)) and the bench with that path
export VPK_PRECISION=1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/cnn_split_trace -o t -- python3 scripts/time_cnn.py --passes 14 102 > $R/cnn_split_trace.log 2>&1
unset VPK_PRECISION
timeout 300 rocprofv3 --kernel-trace --stats -d $R/yud_split_trace -o t -- python3 bench.py --workload yud --steps 20 --warmup 5 --no-cpu-baseline --no-alt --cnn-precision 1 > $R/yud_split_trace.log 2>&1
# MFMA utilisation of the conv / dense kernels: CNN alone and inside the bench (beside the EM)
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d $R/cnn_mfma -o t -- python3 scripts/time_cnn.py --passes 6 102 > $R/cnn_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d $R/yud_mfma -o t -- python3 bench.py --workload yud --steps 4 --warmup 2 --no-cpu-baseline --no-alt > $R/yud_mfma.log 2>&1
python3 scripts/rocpd_stats.py $(find $R/yud_trace -name '*.db' | head -1) $R/summary/${N}_yud_kernel_stats.csv > $R/summary/${N}_yud_top.txt
python3 scripts/rocpd_stats.py $(find $R/stress_trace -name '*.db' | head -1) $R/summary/${N}_stress_kernel_stats.csv > $R/summary/${N}_stress_top.txt
python3 scripts/rocpd_stats.py $(find $R/cnn_trace -name '*.db' | head -1) $R/summary/${N}_cnn_kernel_stats.csv --skip-passes 2 --passes 14 > $R/summary/${N}_cnn_top.txt
python3 scripts/rocpd_stats.py $(find $R/cnn_split_trace -name '*.db' | head -1) $R/summary/${N}_cnn_split_kernel_stats.csv --skip-passes 2 --passes 14 > $R/summary/${N}_cnn_split_top.txt
python3 scripts/rocpd_stats.py $(find $R/cnn_direct_trace -name '*.db' | head -1) $R/summary/${N}_cnn_direct_kernel_stats.csv --skip-passes 2 --passes 14 > $R/summary/${N}_cnn_direct_top.txt
python3 scripts/rocpd_stats.py $(find $R/cnn_wino_trace -name '*.db' | head -1) $R/summary/${N}_cnn_wino_kernel_stats.csv --skip-passes 2 --passes 14 > $R/summary/${N}_cnn_wino_top.txt
python3 scripts/rocpd_stats.py $(find $R/yud_split_trace -name '*.db' | head -1) $R/summary/${N}_yud_split_kernel_stats.csv > $R/summary/${N}_yud_split_top.txt
python3 scripts/rocpd_stats.py $(find $R/cnn_triples_trace -name '*.db' | head -1) $R/summary/${N}_cnn_triples_kernel_stats.csv --skip-passes 2 --passes 14 > $R/summary/${N}_cnn_triples_top.txt
python3 scripts/make_traffic_json.py $R $R/summary/${N}_pmc_traffic.json > /dev/null
python3 scripts/make_mfma_json.py $R $R/summary/${N}_pmc_mfma.json > /dev/null
grep -h '"metric"' $R/yud_trace.log $R/stress_trace.log | cut -c1-400
head -8 $R/summary/${N}_yud_top.txt; head -8 $R/summary/${N}_cnn_top.txt; cat $R/summary/${N}_pmc_mfma.json | head -60
python3 scripts/rocpd_timeline.py $(find $R/yud_trace -name '*.db' | head -1) > $R/summary/${N}_yud_timeline.txt
find $R -name '*.db' -delete
# where the conv2 kernel's wave-cycles go (SQ counters, four passes), the fp16-pair kernel and the bf16-triple one
bash scripts/pmc_kernel.sh conv_pieces_kernelILi5 > $R/summary/${N}_pmc_conv2_issue.txt 2>&1
VPK_ALGORITHM=2 bash scripts/pmc_kernel.sh conv_pieces_kernelILi5 >> $R/summary/${N}_pmc_conv2_issue.txt 2>&1
bash scripts/pmc_kernel.sh conv1_pieces > $R/summary/${N}_pmc_conv1_issue.txt 2>&1
# what the matrix cores sustain with operands that change between instructions
[ -x scripts/ubench/mfma_f16_pairs ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f16_pairs.hip -o scripts/ubench/mfma_f16_pairs > /dev/null 2>&1
./scripts/ubench/mfma_f16_pairs > $R/summary/${N}_mfma_sustained.txt 2>&1
du -sh $R
