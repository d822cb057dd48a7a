#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/mv; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nlist.py -m gpu -x -q -k "lanczos or brownian or overflow or step or sqrt" 2>&1 | tail -4 > $O/pytest.txt
for r in 1 2 3; do timeout 300 python3 tools/perf.py --steps 5 2>&1 | grep "Brownian phases\|step " ; done > $O/perf.txt
cat $O/pytest.txt $O/perf.txt
