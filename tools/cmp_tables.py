#!/usr/bin/env python3
"""Compare two launch tables (bench.py --dump-profile): per (kernel, layer) us per launch, old vs new.
usage: tools/cmp_tables.py old.json new.json [substring]"""
import json, sys
def load(p):
    t = json.load(open(p)); st = t['steps']
    return {(e['kernel'], e['layer']): e['total_ms'] / e['launches'] * 1e3 for e in t['entries']}, \
           {(e['kernel'], e['layer']): e['launches'] / st for e in t['entries']}
a, na = load(sys.argv[1]); b, nb = load(sys.argv[2]); flt = sys.argv[3] if len(sys.argv) > 3 else ''
ta = sum(a[k] * na[k] for k in a); tb = sum(b[k] * nb[k] for k in b)
print(f"sum per step: {ta:.0f} -> {tb:.0f} us")
fam = {}
for k in sorted(set(a) | set(b)):
    f = k[0].split('<')[0]
    fam.setdefault(f, [0, 0]); fam[f][0] += a.get(k, 0) * na.get(k, 0); fam[f][1] += b.get(k, 0) * nb.get(k, 0)
for f, (x, y) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"  {f:28s} {x:8.1f} -> {y:8.1f}")
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, 0))):
    if flt and flt not in k[0]: continue
    if k in a and k in b and abs(a[k] - b[k]) > 1.0:
        print(f"{k[0][:52]:52s} {k[1]:12s} {a[k]:7.1f} -> {b[k]:7.1f}")
