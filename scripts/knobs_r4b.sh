# EM lanes x workgroups per launch at the driver's K = 20 (experiment; the launch lasts as long as its slowest image, so
# more lanes with fewer workgroups each keep the EM's CU share and relax the lane-occupancy bound launch / lanes)
mkdir -p gpurun_out/r4n
for cfg in "3 30" "4 22" "4 26" "5 18" "6 15" "3 30" "4 22"; do
  set -- $cfg
  timeout 200 python bench.py --no-extra --no-cpu-baseline --no-alt --no-from-lines --steps 20 --warmup 5 --em-lanes $1 --em-wgs $2 > gpurun_out/r4n/l$1_w$2.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4n/l$1_w$2.json").read().strip().splitlines()[-1])
print("lanes $1 wgs $2: value %.0f ms/step %.3f cnn %.3f em %.2f" % (d["value"], d["ms_per_step"], d["stage_ms"]["cnn"], d["stage_ms"]["em"]))
PY
done
