#!/usr/bin/env python3
"""Idle time of the main stream inside one training step, from a rocprofv3 kernel trace CSV.

usage: tools/trace_gaps.py <dir-with-*_kernel_trace.csv> [--timeline]
A step is the span between two adam_k launches.  Prints, per step: span, union-busy time, the main stream's own idle
time (gaps between consecutive launches of the stream that carries most kernels), and the ten largest gaps with the
kernels either side -- where a marker packet / stream wait sits in front of a launch.
"""
import csv, glob, re, sys

def short(n):
    n = re.sub(r'^void ', '', n); n = re.sub(r'oct::', '', n); n = re.sub(r'\(.*', '', n)
    return n[:70]

def main():
    d = sys.argv[1]
    files = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)
    if not files:
        print('no *_kernel_trace.csv under', d); return
    import os
    rows = list(csv.DictReader(open(max(files, key=os.path.getmtime))))      # newest run only
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adam_k' in r['Kernel_Name']]
    if len(idx) < 3:
        print('fewer than three steps in the trace'); return
    a, b = idx[-2], idx[-1]
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]['Start_Timestamp']); t1 = int(seg[-1]['End_Timestamp'])
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in seg)
    busy = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: busy += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    busy += ce - cs
    streams = {}
    for r in seg: streams.setdefault(r['Queue_Id'], []).append(r)
    main_q = max(streams, key=lambda q: len(streams[q]))
    m = streams[main_q]
    gaps = []
    for p, n in zip(m, m[1:]):
        g = int(n['Start_Timestamp']) - int(p['End_Timestamp'])
        gaps.append((g, short(p['Kernel_Name']), short(n['Kernel_Name'])))
    print(f"launches {len(seg)}  span {(t1 - t0) / 1e3:.1f} us  busy(union) {busy / 1e3:.1f}  sum of durations {sum(e - s for s, e in iv) / 1e3:.1f}")
    print(f"main queue {main_q}: {len(m)} launches, idle between its launches {sum(g for g, _, _ in gaps if g > 0) / 1e3:.1f} us, "
          f"gaps > 2 us: {sum(1 for g, _, _ in gaps if g > 2000)}")
    for g, p, n in sorted(gaps, reverse=True)[:10]:
        print(f"   {g / 1e3:6.1f} us   {p}  ->  {n}")
    if '--timeline' in sys.argv:
        for r in seg:
            s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
            print(f"q{r['Queue_Id']:>2s} {s / 1e3:8.1f} {e / 1e3:8.1f} {(e - s) / 1e3:7.1f}  {short(r['Kernel_Name'])}")

if __name__ == '__main__':
    main()
