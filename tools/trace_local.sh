#!/bin/bash
# timeline of one owned-particle rank's step (solo mode): every dispatch with its queue, start, duration
export TMPDIR=/tmp
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
O=gpurun_out/local8_trace; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $PYREAL tools/perf_team.py --local --ranks 8 --steps 5 --solo 3 "$@" > $O/trace.log 2>&1
python3 tools/timeline_solo.py $O/trace k_local_classify > $O/solo_timeline.txt 2>&1
grep -E "solo|local team" $O/trace.log; cat $O/solo_timeline.txt
