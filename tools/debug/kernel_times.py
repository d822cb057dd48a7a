#!/usr/bin/env python3
"""Durations of the launches of one kernel in the newest rocprofv3 kernel trace under a directory (developer tool):
  python3 tools/debug/kernel_times.py <dir> <kernel name fragment> [last N launches, default 60]  ->  min / median / max, and the sorted list"""
import csv, glob, os, statistics, sys
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
d = [(int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in csv.DictReader(open(f)) if sys.argv[2] in r['Kernel_Name']]
d = [x[1] for x in sorted(d)][-n:]
print(f"{sys.argv[2]}: {len(d)} launches, min {min(d):.1f} median {statistics.median(d):.1f} max {max(d):.1f} us")
print(' '.join(f"{x:.1f}" for x in d))
