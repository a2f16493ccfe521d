"""one-line summary of a bench.py JSON line"""
import json, sys
for p in sys.argv[1:]:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    wc = d.get("worst_case") or {}
    print(p, d["value"], "ms/step", d["ms_per_step"], r.get("kernels_ms"), "me", r.get("avg_launch_ms"), "tq", (d.get("roofline_tq") or {}).get("avg_launch_ms"),
          "valu", (d.get("valu") or {}).get("frac"), "parity", (d.get("cpu_baseline") or {}).get("parity_with_gpu_on_sample"),
          {k: (v.get("sea_ms"), v.get("brute_ms")) for k, v in wc.items() if isinstance(v, dict)})
