#!/usr/bin/env python3
"""Per-dispatch counters of the 512^3 x pass next to its duration: is the slow mode of an allocation (1.77 against 1.50 ms,
tools/debug/mode_probe.py) more requests, more misses, or the same traffic served more slowly?
  python3 tools/debug/mode_counters.py <rocprofv3 output dir of a --pmc --kernel-trace run of mode_probe.py> [kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d, like = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_xfft_scale_cols")
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print("no counter_collection.csv under", d)
        return
    rows = defaultdict(dict)
    for r in csv.DictReader(open(files[0])):
        if like not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            rows[k]["us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    names = sorted({n for v in rows.values() for n in v if n != "us"})
    print("dispatch     us  " + "  ".join(f"{n:>24s}" for n in names))
    for k in sorted(rows):
        v = rows[k]
        print(f"{k:8d} {v.get('us', float('nan')):7.1f}  " + "  ".join(f"{v.get(n, float('nan')):24.0f}" for n in names))


if __name__ == "__main__":
    main()
