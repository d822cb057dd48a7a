#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: python tools/pmc_table.py <dir-or-csv>... [--like substr]"""
import collections, csv, glob, os, sys
args = [a for a in sys.argv[1:] if not a.startswith("--")]
like = None
if "--like" in sys.argv: like = sys.argv[sys.argv.index("--like") + 1]; args = [a for a in args if a != like]
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for a in args:
    files = [a] if a.endswith(".csv") else glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
            if like and like not in k: continue
            tab[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(tab):
    print(k)
    for c in sorted(tab[k]):
        v = tab[k][c]
        print(f"    {c:42s} {sum(v)/len(v):16.1f}  (n={len(v)})")
