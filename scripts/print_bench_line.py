"""one-line digest of a bench.py JSON line (dev tool): python scripts/print_bench_line.py file.json"""
import json
import sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f img/s  %.3f ms/step  K=%d | fixture %.0f | exact operands %.0f | from lines %.0f | conv2 frac %.3f (alone %.3f) | stream frac %.3f | parity %d/%d"
      % (j["value"], j["ms_per_step"], j["steps"], j.get("value_fixture_prior", 0), j.get("alt_exact_operands", {}).get("value", 0),
         j.get("from_lines", {}).get("value", 0), j["roofline"]["frac"], j["roofline"].get("alone", {}).get("frac", 0), j["cnn_stream"]["frac"],
         j["parity"]["all_criteria"], j["parity"]["images"]))
w = j.get("workloads", {})
if w:
    print("stress %.0f img/s  hlw %.0f img/s" % (w["stress"]["value"], w["hlw"]["images_per_s"]))
print("stage_ms", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in j["stage_ms"].items() if k != "note"})
print("cpu_baseline", {k: j["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind")} if j.get("cpu_baseline") else None)
