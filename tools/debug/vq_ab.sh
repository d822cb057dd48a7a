#!/bin/bash
# PSE_VQ=0 / 1 (the pair-list mat-vec's neighbour rows as doubles / from the 16-byte mirror), alternating fresh processes in one box
R=${1:-3}
for rep in $(seq $R); do
for v in 0 1; do
  echo "== PSE_VQ=$v"
  PSE_VQ=$v python3 bench.py --steps 30 --warmup 10 --no-cpu --no-ref-grid --no-cfg4 --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['phases_ms_per_step']; print('   forked ms_per_step', round(d['ms_per_step'],4), 'm', d['lanczos_m'], {k:p[k] for k in ('lanczos','matvec','real')})"
  PSE_VQ=$v PSE_OVERLAP=-1 python3 bench.py --steps 30 --warmup 10 --no-cpu --no-ref-grid --no-cfg4 --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['phases_ms_per_step']; print('   one stream ms_per_step', round(d['ms_per_step'],4))"
done
done
