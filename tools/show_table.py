#!/usr/bin/env python3
"""Print a launch table written by `bench.py --dump-profile` (per-launch HIP-event times, algorithmic TF and TB/s).
usage: show_table.py <launch_table.json> [substring filter]"""
import json, sys
d = json.load(open(sys.argv[1])); pat = sys.argv[2] if len(sys.argv) > 2 else ""
n = d["steps"]; tot = 0.0; fam = {}
for e in d["entries"]:
    ms = e["total_ms"] / n; tot += ms
    f = fam.setdefault(e["kernel"].split("<")[0], [0.0, 0.0, 0.0]); f[0] += ms; f[1] += e["flops"] / n; f[2] += e["bytes"] / n
    if pat and pat not in e["kernel"]:
        continue
    if pat:
        print(f"{e['layer']:12s} {e['kernel'][:46]:46s} {ms*1000:8.1f}us {e['flops']/n/ms/1e9 if ms else 0:7.1f}TF {e['bytes']/n/ms/1e9 if ms else 0:6.2f}TB/s")
print(f"total {tot:.3f} ms/step")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:22s} {v[0]*1000:8.1f}us {v[1]/v[0]/1e9:7.1f}TF {v[2]/v[0]/1e9:6.2f}TB/s")
