mkdir -p gpurun_out/r4c
for w in 22 26 30 34 40; do
  timeout 200 python bench.py --no-extra --no-cpu-baseline --no-alt --steps 20 --em-wgs $w > gpurun_out/r4c/wgs_$w.json 2>/dev/null
done
for l in 2 4; do
  timeout 200 python bench.py --no-extra --no-cpu-baseline --no-alt --steps 20 --em-lanes $l > gpurun_out/r4c/lanes_$l.json 2>/dev/null
done
for l in 2 4; do
  timeout 200 python bench.py --no-extra --no-cpu-baseline --no-alt --steps 20 --em-lanes $l --em-wgs 40 > gpurun_out/r4c/lanes_${l}_w40.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['stage_ms']['cnn'],3), round(d['stage_ms']['em'],2))
    except Exception as e: print(f, 'ERR', e)
PY
