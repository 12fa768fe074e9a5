#!/bin/bash
# time-sliced EM launches with the fp16-pair CNN (2.7 ms alone): workgroups x slice budget (dev tool)
mkdir -p gpurun_out/r5c; rm -f gpurun_out/r5c/*.json
for cfg in "96 3" "112 3" "128 3" "128 2.5" "144 3" "144 2.5" "160 2.5" "160 3" "176 2.5" "128 3.5" "112 3.5"; do
  set -- $cfg
  timeout 200 python bench.py --no-extra --no-cpu-baseline --no-alt --steps 20 --warmup 5 --em-mode slice --em-wgs $1 --em-slice-ms $2 > gpurun_out/r5c/slice_w$1_t$2.json 2>/dev/null
done
timeout 200 python bench.py --no-extra --no-cpu-baseline --no-alt --steps 20 --warmup 5 --em-mode lanes > gpurun_out/r5c/lanes.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5c/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3), {k: (round(v,2) if isinstance(v,float) else v) for k,v in d['stage_ms'].items() if k not in ('note','em_mode')}, d['parity']['all_criteria'])
    except Exception as e: print(f, 'ERR', e)
PY
