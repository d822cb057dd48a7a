#!/bin/bash
# parity of the Brownian paths, then the step, the phases and the mat-vec time
python -m pytest tests/test_gpu_parity.py tests/test_gpu_nlist.py -m gpu -x -q -k "brownian or lanczos or step or mreal or pair_list or overflow or reused" 2>&1 | tail -3
for rep in 1 2 3; do
  python3 bench.py --steps 30 --warmup 10 --no-cpu --no-ref-grid --no-cfg4 --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['phases_ms_per_step']; r=d['roofline']; print('   forked ms_per_step', round(d['ms_per_step'],4), 'm', d['lanczos_m'], {k:p[k] for k in ('lanczos','matvec','real')}, r['ms_per_launch'], r['ms_per_launch_back_to_back_warm'], d['grid_placement']['ms_kept'])"
  PSE_OVERLAP=-1 python3 bench.py --steps 30 --warmup 10 --no-cpu --no-ref-grid --no-cfg4 --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   one stream ms_per_step', round(d['ms_per_step'],4))"
done
