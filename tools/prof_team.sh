#!/bin/bash
# Eight-rank loopback team on the one-GPU box: the team tests, per-rank kernel times with every kernel ALONE on one stream
# (rocprofv3 --stats of tools/perf_team.py with PSE_OVERLAP=-1), the team step with the two lanes overlapping (the default), and
# one rank's critical path with the GPU to itself (tools/perf_team.py --solo: both lanes, its share of the copies) with its timeline.
export TMPDIR=/tmp
# the interpreter itself after `--` (a launcher that re-execs under the profiler is refused on the GPU box)
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
O=gpurun_out/team8; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_fullsize.py -m gpu -x -q -k "slab or team or loopback or rccl or lanczos" 2>&1 | grep -E "passed|failed|rror" > $O/pytest.txt
timeout 300 python3 tools/perf_team.py --ranks 8 --steps 5 2>&1 | grep -E "team of|phases|Lanczos" > $O/perf_team8.txt
PSE_OVERLAP=-1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $PYREAL tools/perf_team.py --ranks 8 --steps 5 > $O/stats.log 2>&1
python3 - $O <<'PY' > $O/team8_kernel_stats.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("# PSE_OVERLAP=-1 (one stream: every kernel alone) rocprofv3 --kernel-trace --stats -- tools/perf_team.py --ranks 8 --steps 5; per call = per rank")
print("total kernel ms", tot/1e6)
for r in rows[:40]:
    print(f"{r['Name'].split('(')[0].replace('void ','')[:60]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}")
PY
{
for cfg in "PSE_TEAM_SSTEP=1" "PSE_TEAM_SSTEP=0" "PSE_TEAM_SSTEP=1 PSE_OVERLAP=-1" "PSE_TEAM_SSTEP=0 PSE_OVERLAP=-1"; do
  echo "== $cfg"; env $cfg timeout 300 python3 tools/perf_team.py --ranks 8 --steps 30 --solo 3 2>&1 | grep -E "solo|team of"
done
} > $O/solo.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $PYREAL tools/perf_team.py --ranks 8 --steps 20 --solo 3 > $O/trace.log 2>&1
python3 tools/timeline_solo.py $O/trace > $O/solo_timeline.txt 2>&1
cat $O/pytest.txt $O/perf_team8.txt; head -44 $O/team8_kernel_stats.txt; cat $O/solo.txt; tail -4 $O/solo_timeline.txt
