#!/bin/bash
# shader clock and matrix-pipe utilisation per CNN kernel (CNN alone): one --pmc pass (dev tool)   VPK_ALGORITHM=2 bash scripts/pmc_cnn_clock.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/pmc_clock
rm -rf $R; mkdir -p $R
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $R/cnn_mfma -o t -- python3 scripts/time_cnn.py --passes 5 102 > $R/run.log 2>&1
python3 - <<'PY'
import sys, os
sys.path.insert(0, "scripts")
from make_traffic_json import first_db, per_kernel
k = per_kernel(first_db("gpurun_out/pmc_clock/cnn_mfma"))
for name, c in sorted(k.items(), key=lambda kv: -kv[1]["_ms"])[:9]:
    gui, busy = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    print("%-60s %.3f ms  clock %.2f GHz  mfma busy %.2f  wait %.2f stall %.2f" % (name[24:84], c["_ms"], gui / 8 / (c["_ms"] * 1e6) if gui else 0,
          busy / (gui / 8 * 1024) if gui else 0, c.get("SQ_WAIT_ANY", 0) / (c.get("SQ_WAVE_CYCLES", 1) or 1), c.get("SQ_WAIT_INST_ANY", 0) / (c.get("SQ_WAVE_CYCLES", 1) or 1)))
PY
find $R -name '*.db' -delete
