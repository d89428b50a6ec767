"""one-line digest of a bench.py JSON line read from stdin (development aid)"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads([ln for ln in sys.stdin.read().splitlines() if ln.startswith("{")][-1])
r = d["roofline"]
print(tag, round(d["value"]), round(d["ms_per_step"], 4), "dft", round(d.get("dft_kernel", {}).get("avg_launch_us", 0), 1),
      [(c["workgroups"], round(c["avg_us"], 1)) for c in r["launch_classes"]], "frac", round(r["frac"], 3))
