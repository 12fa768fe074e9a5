#!/bin/bash
# EM lanes x workgroups sweep of the headline run (dev tool)
mkdir -p gpurun_out/r5k
for cfg in "3 30" "3 36" "4 22" "4 26" "4 30" "5 20" "5 24"; do
  set -- $cfg
  timeout 200 python bench.py --no-extra --no-cpu-baseline --no-alt --steps 20 --warmup 5 --em-lanes $1 --em-wgs $2 > gpurun_out/r5k/l$1_w$2.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5k/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['stage_ms']['cnn'],3), round(d['stage_ms']['em'],2))
    except Exception as e: print(f, 'ERR', e)
PY
