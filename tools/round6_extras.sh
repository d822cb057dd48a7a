#!/bin/bash
# Round-6 evidence beside tools/round_profile.sh (tools/round6_extras.sh <tag>): phase timings of the BASELINE configurations,
# the per-rank critical path of eight-rank teams (owned-particle step and replicated-state step, metric point and config 4) from
# pse_team_debug_solo, the timeline of one owned-particle rank's step, a two-rank bench line over the host-staged transport with
# its per-exchange diagnosis.  Output under gpurun_out/<tag>/.
export TMPDIR=/tmp
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
T=${1:-r05x}
ONLY=${2:-all}     # "team": the team measurements only (the configurations take most of the time)
O=gpurun_out/$T; mkdir -p $O
if [ "$ONLY" = all ]; then
{
for a in "--n 1000 --phi 0.05 --grid 64 --only-mf --steps 200" "--n 65536 --grid 64 --only-mf --steps 200" "--n 1048576 --phi 0.2 --grid 256 --steps 10" "--n 1048576 --phi 0.1 --grid 256 --xy 0.3 --steps 10" "--n 4194304 --phi 0.3 --grid 512 --steps 5" "--grid 0 --xi 0.5 --steps 40"; do
  echo "== tools/perf.py $a"; timeout 600 python3 tools/perf.py $a 2>&1 | grep -E "create|phases|M.F |^step|queue-only"
done
} > $O/configs.txt
fi
{
echo "# one rank's critical path with the GPU to itself (pse_team_debug_solo), eight ranks, zero-latency links (device copies)"
echo "# extra 0 = the steady state of a time-stepping loop (pse_team_set_lanczos_extra(0): no gated block queued, what bench.py times); extra -1 = the gated block kept"
for ex in 0 -1; do
echo "== owned-particle step (pse_team_step_local), metric point N = 1e6, 256^3, TWO LANES, extra $ex"
timeout 600 python3 tools/perf_team.py --local --ranks 8 --solo 3 --extra $ex 2>&1 | grep -E "layout|local team|solo"
echo "== the same on ONE STREAM (PSE_TEAM_LANES=0), extra $ex"
PSE_TEAM_LANES=0 timeout 600 python3 tools/perf_team.py --local --ranks 8 --solo 3 --extra $ex 2>&1 | grep -E "solo rank 3"
done
for ex in 0 -1; do
echo "== owned-particle step, BASELINE config 4 (N = 4194304, phi = 0.3, 512^3), TWO LANES, extra $ex"
timeout 600 python3 tools/perf_team.py --local --ranks 8 --solo 3 --extra $ex --n 4194304 --phi 0.3 --grid 512 --steps 3 2>&1 | grep -E "layout|local team|solo"
echo "== the same on ONE STREAM, extra $ex"
PSE_TEAM_LANES=0 timeout 600 python3 tools/perf_team.py --local --ranks 8 --solo 3 --extra $ex --n 4194304 --phi 0.3 --grid 512 --steps 3 2>&1 | grep -E "solo rank 3"
done
echo "== owned-particle step, four ranks, metric point, two lanes, extra 0"
timeout 600 python3 tools/perf_team.py --local --ranks 4 --solo 1 --extra 0 2>&1 | grep -E "local team|solo rank 1 of|back to back"
echo "== replicated-state step (pse_team_step; Brownian evaluation without the Euler update), metric point"
timeout 600 python3 tools/perf_team.py --ranks 8 --solo 3 2>&1 | grep -E "team of|solo"
echo "== the two halves of the two-rank functional split (tools/perf_split.py), metric point and config 4"
timeout 600 python3 tools/perf_split.py 2>&1 | tail -8
timeout 600 python3 tools/perf_split.py --n 4194304 --phi 0.3 --grid 512 --steps 5 2>&1 | tail -8
} > $O/team8_solo_times.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- $PYREAL tools/perf_team.py --local --ranks 8 --steps 5 --solo 3 --extra 0 > $O/trace.log 2>&1
{ echo "# one EAGER solo step of rank 3 of 8 (owned-particle step, metric point) under rocprofv3 --kernel-trace: tools/timeline_solo.py <trace> k_local_classify -50"
  echo "# (the last steps of the run are replayed hipGraphs, whose nodes the runtime launches branch by branch: not this timeline)"
  python3 tools/timeline_solo.py $O/trace k_local_classify -50; } > $O/team8_local_timeline.txt 2>&1
PSE_TEAM_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- $PYREAL tools/perf_team.py --local --ranks 8 --steps 5 --solo 3 --extra 0 > $O/stats1.log 2>&1
python3 - $O <<'PY' > $O/team8_local_kernel_stats.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/stats1/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("# PSE_TEAM_LANES=0 (one stream: every kernel alone) rocprofv3 --kernel-trace --stats -- tools/perf_team.py --local --ranks 8 --steps 5 --solo 3 --extra 0")
print("# (full team steps of all eight ranks and solo steps of rank 3 in one run: average durations are per rank-launch)")
for r in rows[:40]:
    print(f"{r['Name'].split('(')[0].replace('void ','')[:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}")
PY
timeout 900 python3 bench.py --gpus 2 --transport host --steps 10 --warmup 5 --no-cpu > $O/bench_2ranks_host.json 2> $O/bench_2ranks_host.err
tail -3 $O/configs.txt; cat $O/team8_solo_times.txt; tail -5 $O/team8_local_timeline.txt; tail -c 900 $O/bench_2ranks_host.json
