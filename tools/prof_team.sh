#!/bin/bash
# Eight-rank loopback team on the one-GPU box: the team tests, then per-rank kernel times (rocprofv3 --stats of tools/perf_team.py)
export TMPDIR=/tmp
# the interpreter itself after `--` (a launcher that re-execs under the profiler is refused on the GPU box)
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
O=gpurun_out/team8; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_fullsize.py -m gpu -x -q -k "slab or team or loopback or rccl" 2>&1 | tail -8 > $O/pytest.txt
timeout 300 python3 tools/perf_team.py --ranks 8 --steps 5 > $O/perf_team8.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $PYREAL tools/perf_team.py --ranks 8 --steps 5 > $O/stats.log 2>&1
python3 - $O <<'PY' > $O/team8_kernel_stats.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:36]:
    print(f"{r['Name'].split('(')[0].replace('void ','')[:60]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}")
PY
cat $O/pytest.txt; tail -4 $O/perf_team8.txt; head -40 $O/team8_kernel_stats.txt
