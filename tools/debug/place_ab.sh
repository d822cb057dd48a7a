#!/bin/bash
# The grid-placement planner of pse_create (place_grids, PSE_PLACE_TRIALS) off / on, alternating, fresh processes in ONE box:
# step time, far-field phases, the candidates' probe times, and what the planner costs at create.
#   bash tools/debug/place_ab.sh [repeats] [extra bench arguments, e.g. "--n 4194304 --phi 0.3 --grid 512 --steps 8 --warmup 3"]
mkdir -p gpurun_out/place
R=${1:-4}; X=${2:-}
for rep in $(seq $R); do
for k in 0 6; do
  echo "== PSE_PLACE_TRIALS=$k"
  PSE_PLACE_TRIALS=$k PSE_VERBOSE=1 python3 bench.py --steps 30 --warmup 10 --no-cpu --no-ref-grid --no-cfg4 --no-traffic $X 2>gpurun_out/place/err_${k}_${rep}.txt | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['phases_ms_per_step']; print('   ms_per_step', round(d['ms_per_step'],4), 'mf', round(d['mf_evals_per_s'],1), round(d['mf_evals_per_s_moving'],1), {k:p[k] for k in ('fft_fwd','scale','fft_inv')})"
  grep "grid placement" gpurun_out/place/err_${k}_${rep}.txt | cut -c1-120
done
done
