#!/bin/bash
# time-sliced EM launches against the lanes scheme with the round-5 CNN (dev tool)
mkdir -p gpurun_out/r5s; rm -f gpurun_out/r5s/*.json
for cfg in "96 4" "96 3" "112 4" "112 3" "128 3" "128 4" "104 3.5" "96 5" "144 3"; do
  set -- $cfg
  timeout 200 python bench.py --no-extra --no-cpu-baseline --no-alt --steps 20 --warmup 5 --em-mode slice --em-wgs $1 --em-slice-ms $2 > gpurun_out/r5s/slice_w$1_t$2.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5s/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3), {k: (round(v,2) if isinstance(v,float) else v) for k,v in d['stage_ms'].items() if k not in ('note','em_mode')}, d['parity']['all_criteria'])
    except Exception as e: print(f, 'ERR', e)
PY
