import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"],4), "ms/step", round(d["ms_per_step"],1), "kernel us", round(r["us_per_launch"],2), "frac", round(r["frac"],3), "decode avg ms", round(r["decode_avg_step_ms"],3), "avg frac", round(r["decode_avg_frac"],3), "prefill", round(r["prefill_ms"],2))
if "b32" in d: b=d["b32"]; print("b32", round(b["value"],2), round(b["decode_avg_step_ms"],3), round(b["decode_avg_frac"],3), "prefill", round(b["prefill_ms"],1))
