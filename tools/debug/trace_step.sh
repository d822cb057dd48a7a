#!/bin/bash
# timeline of one single-GPU Brownian step of the bench (developer tool): every dispatch with its queue, start, duration, idle time
export TMPDIR=/tmp
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
O=gpurun_out/step_trace; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $PYREAL tools/perf.py --n 1000000 --phi 0.1 --grid 256 --steps 12 "$@" > $O/trace.log 2>&1
python3 tools/timeline_solo.py $O/trace k_cell_keys -6 > $O/timeline.txt 2>&1
tail -5 $O/trace.log; cat $O/timeline.txt
