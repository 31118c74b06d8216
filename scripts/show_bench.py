"""One screen of a bench.py line: python scripts/show_bench.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"])
k = d.get("kernels") or d.get("config", {}).get("kernels")
for key in ("kernel_ms_per_step", "kernels", "per_step"):
    if key in d: print(key, d[key])
r = d.get("roofline", {})
print("roofline:", {k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "avg_launch_ms", "launches")})
if "k_shade" in r: print("k_shade:", r["k_shade"])
for name, v in (d.get("variants") or {}).items():
    if name == "other_configs":
        for c, w in v.items(): print(" ", c, {k: w.get(k) for k in ("value", "ms", "k_extend_ms", "k_shade_ms", "segments")})
    else: print(name, {k: v.get(k) for k in ("value", "ms_per_step", "extend_ms", "error")} if isinstance(v, dict) else v)
