import sys, json
d = json.loads(sys.stdin.read())
r = d["roofline"]
print(d["ms_per_step"], d["value"], r["kernel_ms"], r.get("kernel_ms_in_flight"))
