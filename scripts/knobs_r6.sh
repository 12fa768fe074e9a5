#!/bin/bash
# time-sliced EM launches with the fp16-pair CNN (2.5 ms alone, round 6): workgroups x slice budget, 30 steps like the default run (dev tool)
mkdir -p gpurun_out/r6k; rm -f gpurun_out/r6k/*.json
for cfg in "160 2.5" "144 2.5" "176 2.5" "160 2.25" "176 2.25" "144 2.25" "160 2.0" "192 2.0" "160 2.5" "128 2.75" "176 2.0"; do
  set -- $cfg
  timeout 200 python bench.py --no-extra --no-cpu-baseline --no-alt --steps 30 --warmup 4 --em-mode slice --em-wgs $1 --em-slice-ms $2 > gpurun_out/r6k/slice_w$1_t$2.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6k/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3), {k: (round(v,2) if isinstance(v,float) else v) for k,v in d['stage_ms'].items() if k not in ('note','em_mode')}, d['parity']['all_criteria'])
    except Exception as e: print(f, 'ERR', e)
PY
