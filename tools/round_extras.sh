#!/bin/bash
# Extra round evidence on the GPU box (tools/round_extras.sh <tag>): the kernel timeline of one overlapped step, the phase timings of the
# other BASELINE configurations, the per-rank kernel times of an 8-slab loopback team.  Output under gpurun_out/<tag>/.
export TMPDIR=/tmp
# the interpreter itself after `--` (a launcher that re-execs under the profiler is refused on the GPU box)
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
T=${1:-r02x}
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $PYREAL tools/perf.py --steps 6 > $O/trace.log 2>&1
python3 tools/timeline.py $O/trace > $O/overlap_timeline.txt 2>&1
{
for a in "--n 65536 --grid 64 --only-mf --steps 200" "--n 1048576 --phi 0.2 --grid 256 --steps 10" "--n 1048576 --phi 0.1 --grid 256 --xy 0.3 --steps 10" "--n 4194304 --phi 0.3 --grid 512 --steps 5" "--grid 0 --xi 0.5 --steps 5"; do
  echo "== tools/perf.py $a"; timeout 600 python3 tools/perf.py $a 2>&1 | grep -E "create|phases|M.F |^step"
done
} > $O/configs.txt
bash tools/prof_team.sh > $O/team8.txt 2>&1
tail -3 $O/configs.txt
