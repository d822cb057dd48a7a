#!/bin/bash
# timeline of one deterministic evaluation at BASELINE config 2 (N = 65536, 64^3): every dispatch with its queue, start, duration
export TMPDIR=/tmp
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
O=gpurun_out/cfg2_trace; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $PYREAL tools/perf.py --n 65536 --grid 64 --only-mf --steps 50 > $O/trace.log 2>&1
python3 tools/timeline_solo.py $O/trace k_permute > $O/cfg2_timeline.txt 2>&1
grep -E "M.F " $O/trace.log; cat $O/cfg2_timeline.txt
