#!/usr/bin/env python3
"""Per-layer view of a bench.py --dump-profile launch table (us per step, MB and GF per launch).
usage: tools/show_layers.py table.json [other.json]  (two tables: side by side per layer totals)"""
import json, sys

def load(p):
    d = json.load(open(p)); steps = d["steps"]
    by = {}
    for e in d["entries"]:
        by.setdefault(e["layer"], []).append((e["kernel"], e["total_ms"] * 1000 / steps, e["launches"] // steps, e["bytes"] / steps / 1e6, e["flops"] / steps / 1e9))
    return by

a = load(sys.argv[1]); b = load(sys.argv[2]) if len(sys.argv) > 2 else None
tot = 0
for layer, es in a.items():
    t = sum(x[1] for x in es); tot += t
    print(f"--- {layer}: {t:.0f} us" + (f"   (other: {sum(x[1] for x in b.get(layer, [])):.0f})" if b else ""))
    for k, us, n, mb, gf in es:
        print(f"   {us:8.1f} us x{n}  {k:52s} {mb:6.0f} MB {mb / us / 1e3 if us else 0:5.2f} TB/s {gf:5.1f} GF {gf / us * 1e3 if us else 0:6.1f} TF")
print(f"total {tot:.0f} us" + (f"  (other {sum(x[1] for es in b.values() for x in es):.0f})" if b else ""))
