#!/bin/bash
# Paired A/B runs of one environment switch in ONE box (the boxes of the pool differ by +-2 %): alternates the two settings.
#   bash tools/debug/solo_ab.sh "PSE_OVERLAP=0" "PSE_OVERLAP=1" [repeats]       headline bench: ms per step, M.F evals/s (fixed / moving), phases
#   TEAM=1 bash tools/debug/solo_ab.sh "PSE_SIDE_PRIORITY=default" "PSE_SIDE_PRIORITY=low"   solo rank of an eight-rank owned-particle team
A=${1:-PSE_OVERLAP=0}; Bv=${2:-PSE_OVERLAP=1}; R=${3:-2}
for rep in $(seq $R); do
  for v in "$A" "$Bv"; do
    echo "== $v"
    if [ -n "$TEAM" ]; then
      env $v python3 tools/perf_team.py --local --ranks 8 --steps 5 --solo 3 2>&1 | grep -E "solo"
    else
      env $v python3 bench.py --steps 30 --warmup 10 --no-cpu --no-ref-grid --no-cfg4 --no-traffic 2>&1 | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['phases_ms_per_step']; print(round(d['ms_per_step'],4), round(d['mf_evals_per_s'],1), round(d['mf_evals_per_s_moving'],1), {k:p[k] for k in ('sort','real','matvec','lanczos','spread','gather')})"
    fi
  done
done
