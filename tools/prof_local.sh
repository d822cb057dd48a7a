#!/bin/bash
# Owned-particle team of eight on the one-GPU box: per-kernel times of ONE rank's step (solo mode, rocprofv3 --kernel-trace --stats)
# usage: tools/prof_local.sh [extra perf_team.py args]
export TMPDIR=/tmp
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
O=gpurun_out/local8; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $PYREAL tools/perf_team.py --local --ranks 8 --steps 5 --solo 3 "$@" > $O/stats.log 2>&1
python3 - $O <<'PY' > $O/local8_kernel_stats.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:45]:
    print(f"{r['Name'].split('(')[0].replace('void ','')[:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}")
PY
tail -5 $O/stats.log; cat $O/local8_kernel_stats.txt
