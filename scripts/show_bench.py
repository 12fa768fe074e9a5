"""Print the headline fields of a bench.py JSON line (dev tool):  python scripts/show_bench.py gpurun_out/x/bench.json"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "steps", "warmup", "value_fixture_prior")})
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["achieved"], 1), round(d["roofline"]["frac"], 3), d["stage_ms"])
print("layers", d.get("cnn_layer_ms"))
for k in ("alt_precision", "from_lines"):
    if k in d:
        print(k, round(d[k]["value"]), round(d[k]["ms_per_step"], 2))
print({k: (v.get("value"), v.get("ms_per_step")) for k, v in d.get("workloads", {}).items()})
if "parity" in d:
    print("parity", d["parity"]["all_criteria"], d["parity"]["rasters_equal_reference"])
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"])
