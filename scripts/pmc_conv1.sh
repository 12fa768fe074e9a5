#!/bin/bash
# SQ counters of the CNN kernels alone (one --pmc set per pass, --kernel-trace only): LDS activity / conflicts / waits.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/pmc1
rm -rf $R; mkdir -p $R
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $R/a -o t --output-format csv -- python3 scripts/time_cnn.py --passes 4 102 > $R/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU -d $R/b -o t --output-format csv -- python3 scripts/time_cnn.py --passes 4 102 > $R/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob("gpurun_out/pmc1/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            if "conv1_direct" in k or "conv_gemm" in k:
                print(k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
find $R -name '*.db' -delete
