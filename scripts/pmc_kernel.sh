#!/bin/bash
# where a CNN kernel's wave-cycles go: four --pmc passes over the CNN alone, per-kernel ratios (dev tool)
#   bash scripts/pmc_kernel.sh [kernel-name substring, default conv_pieces]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=${1:-conv_pieces}
R=gpurun_out/pmc_kernel
rm -rf $R; mkdir -p $R
run() { timeout 300 rocprofv3 --kernel-trace --pmc $2 -d $R/$1 -o t -- python3 scripts/time_cnn.py --passes 4 102 > $R/$1.log 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES"
run c "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_UNALIGNED_STALL"
run d "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
python3 - "$K" <<'PY'
import sys
sys.path.insert(0, "scripts")
from make_traffic_json import first_db, per_kernel
tot = {}
for p in "abcd":
    try:
        k = per_kernel(first_db("gpurun_out/pmc_kernel/" + p))
    except Exception as e:
        print("pass", p, "failed:", e); continue
    for name, c in k.items():
        if sys.argv[1] in name:
            tot.setdefault(name, {}).update(c)
for name, c in tot.items():
    print(name[:100])
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    for key in sorted(c):
        if key.startswith("_"): print("   %-32s %s" % (key, c[key])); continue
        print("   %-32s %14.0f   /wave-cycles %.4f" % (key, c[key], c[key] / wc))
PY
find $R -name '*.db' -delete
