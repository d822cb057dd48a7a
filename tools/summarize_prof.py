#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

  python tools/summarize_prof.py stats  <kernel_stats.csv> <out.csv> "<command line that was profiled>"
  python tools/summarize_prof.py pmc    <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [commit] [command]
"""
import collections
import csv
import json
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")[:80]


def stats(src, dst, cmd):
    """The profiler's per-kernel averages, and beside them the MEDIAN duration from the kernel trace of the same run: the averages of the
    inverse y and z passes include the launches of pse_create's grid-placement probe (four per candidate pair, some on slow pairs)."""
    import os
    import statistics
    rows = list(csv.DictReader(open(src)))
    med = {}
    trace = src.replace("kernel_stats", "kernel_trace")
    if trace != src and os.path.exists(trace):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(trace)):
            d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        med = {k: statistics.median(v) for k, v in d.items()}
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n")
        f.write("# MedianNs: from the kernel trace of the same run (the averages of k_xfft_scale_cols, k_yfft_regs<..true..> and k_zfft_rows<..true..> include the launches of the create-time placement probe (four per candidate pair))\n")
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MedianNs\n")
        for r in rows[:32]:
            m = med.get(r["Name"])
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{'' if m is None else int(m)}\n")


def pmc(fetch_csv, write_csv, dst, commit="unrecorded", cmd=""):
    def agg(path, counter):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in d.items()}
    f, w = agg(fetch_csv, "FETCH_SIZE"), agg(write_csv, "WRITE_SIZE")
    out = {"_note": "per-launch averages, bytes. FETCH_SIZE/WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a 128-B "
                    "request as 64 B for wide (16 B/lane) streaming reads, so fetch_x2 is the corrected figure for kernels that read with "
                    "dwordx4 (/opt/skills/guides/MI355X_MICROARCH.md, HBM section); calibrated in round 3 for plain and non-temporal 8- and "
                    "16-byte-per-lane streams and for the ragged pair-list stream of k_mreal_list (tools/microbench/nt_fetch.hip, "
                    "profiles/r03_fetch_size_calibration.txt: x2 in every case, 96 % of the lines touched for the list)."}
    out["_commit"] = commit
    out["_source"] = "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of: " + cmd
    for k in sorted(set(f) | set(w)):
        out[k] = {"fetch_raw": f.get(k, 0.0) * 1024, "fetch_x2": 2 * f.get(k, 0.0) * 1024, "write": w.get(k, 0.0) * 1024}
    json.dump(out, open(dst, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], *(sys.argv[5:7]))
