import sys, json
d = json.loads(sys.stdin.read())
print(d["ms_per_step"], {k: v.get("ms_per_step", v.get("error")) for k, v in d.items() if k.startswith("secondary")})
