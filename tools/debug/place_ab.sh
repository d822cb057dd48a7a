#!/bin/bash
# The grid-placement planner of pse_create (place_grids, PSE_PLACE_TRIALS) off / on, alternating, fresh processes in ONE box:
# step time, far-field phases, the candidates' probe times, and what the planner costs at create.   bash tools/debug/place_ab.sh [repeats]
mkdir -p gpurun_out/place
R=${1:-4}
for rep in $(seq $R); do
for k in 0 6; do
  echo "== PSE_PLACE_TRIALS=$k"
  PSE_PLACE_TRIALS=$k PSE_VERBOSE=1 python3 bench.py --steps 30 --warmup 10 --no-cpu --no-ref-grid --no-cfg4 --no-traffic 2>gpurun_out/place/err_${k}_${rep}.txt | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['phases_ms_per_step']; print('   ms_per_step', round(d['ms_per_step'],4), 'mf', round(d['mf_evals_per_s'],1), round(d['mf_evals_per_s_moving'],1), {k:p[k] for k in ('fft_fwd','scale','fft_inv')})"
  grep "grid placement" gpurun_out/place/err_${k}_${rep}.txt | cut -c1-120
done
done
python3 - <<'PY'
import os, sys, time, math
sys.path.insert(0, os.getcwd())
import torch, pse_amd
torch.zeros(1, device="cuda")
L = 161.2
for k in (0, 6, 0, 6):
    os.environ["PSE_PLACE_TRIALS"] = str(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = pse_amd.Engine(1_000_000, (L, L, L, 0.0), xi=math.pi * 256 / (2.0 * L * math.sqrt(-math.log(1e-3))), error=1e-3, seed=1, grid=(256, 256, 256))
    torch.cuda.synchronize(); print("pse_create with PSE_PLACE_TRIALS=%d: %.1f ms" % (k, (time.perf_counter() - t0) * 1e3))
    del e
PY
