#!/usr/bin/env python3
"""Timeline of ONE rank's Brownian evaluation from a rocprofv3 kernel trace of `tools/perf_team.py --solo R` (developer tool):
every dispatch between the last two k_cell_keys launches with its queue, start, duration and the idle time of its queue before
it, then per queue: busy time, idle time between its first and last dispatch; and how long two queues were busy at once.
  python3 tools/timeline_solo.py <dir with *_kernel_trace.csv> [marker kernel, default k_cell_keys] [index of the step's marker, default -2]"""
import csv
import glob
import sys

import os
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)   # (merged run directories accumulate: the newest)
marker = sys.argv[2] if len(sys.argv) > 2 else 'k_cell_keys'
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
k = int(sys.argv[3]) if len(sys.argv) > 3 else -2
a, b = marks[k], marks[k + 1]
# the call starts with the memset in front of the marker kernel
while a > 0 and rows[a - 1]['s'] > rows[a]['s'] - 20000 and 'fillBuffer' in rows[a - 1]['Kernel_Name']:
    a -= 1
while b > a and rows[b - 1]['s'] > rows[b]['s'] - 20000 and 'fillBuffer' in rows[b - 1]['Kernel_Name']:
    b -= 1
st = rows[a:b]
t0 = st[0]['s']
last_end, busy = {}, {}
for r in st:
    q = r['Queue_Id']
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('pse::', '')[:52]
    gap = (r['s'] - last_end[q]) / 1e3 if q in last_end else 0.0
    print(f"q{q:>2s} {(r['s'] - t0) / 1e3:9.1f} +{(r['e'] - r['s']) / 1e3:7.1f} us  idle before {gap:6.1f}  {nm}")
    last_end[q] = max(last_end.get(q, 0), r['e'])
    busy.setdefault(q, []).append((r['s'], r['e']))
span = (max(r['e'] for r in st) - t0) / 1e3
print(f"span {span:.1f} us, {len(st)} dispatches")
for q, iv in busy.items():
    b_ = sum(e - s for s, e in iv) / 1e3
    print(f"queue {q}: {len(iv)} dispatches, busy {b_:.1f} us, idle between its first and last dispatch {(iv[-1][1] - iv[0][0]) / 1e3 - b_:.1f} us")
ev = sorted([(s, 1) for iv in busy.values() for s, e in iv] + [(e, -1) for iv in busy.values() for s, e in iv])
d, last, both, any_ = 0, t0, 0, 0
for t, k in ev:
    if d >= 2: both += t - last
    if d >= 1: any_ += t - last
    d += k; last = t
print(f"some queue busy {any_ / 1e3:.1f} us, two or more at once {both / 1e3:.1f} us, nothing running {span - any_ / 1e3:.1f} us")
