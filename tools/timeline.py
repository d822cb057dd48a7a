#!/usr/bin/env python3
"""Timeline of the last step in a rocprofv3 kernel trace (developer tool): queue, start, duration of every dispatch between
the last two k_integrate launches, and how much of the step two queues were busy at once.
  python3 tools/timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
for r in rows: r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ends = [i for i, r in enumerate(rows) if 'k_integrate' in r['Kernel_Name']]
a, b = ends[-2] + 1, ends[-1] + 1
st = rows[a:b]
t0 = st[0]['s']
busy = {}
for r in st:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('pse::', '')[:58]
    print(f"q{r['Queue_Id']:>2s} {(r['s']-t0)/1e3:9.1f} +{(r['e']-r['s'])/1e3:8.1f} us  {nm}")
    busy.setdefault(r['Queue_Id'], []).append((r['s'], r['e']))
print("step span %.1f us" % ((st[-1]['e'] - t0) / 1e3))
for q, iv in busy.items(): print("queue", q, "busy %.1f us" % (sum(e - s for s, e in iv) / 1e3))
ev = sorted([(s, 1) for iv in busy.values() for s, e in iv] + [(e, -1) for iv in busy.values() for s, e in iv])
d, last, both, any_ = 0, t0, 0, 0
for t, k in ev:
    if d >= 2: both += t - last
    if d >= 1: any_ += t - last
    d += k; last = t
print("any busy %.1f us, two or more at once %.1f us" % (any_ / 1e3, both / 1e3))
