# A/B runs of the lane order / lane priority (developer scratch: edit freely)
B="python bench.py --steps 30 --warmup 8 --no-cpu --no-ref-grid --no-traffic"
for v in "PSE_OVERLAP=0" "PSE_OVERLAP=1" "PSE_OVERLAP=1 PSE_SIDE_PRIORITY=low" "PSE_OVERLAP=0" "PSE_OVERLAP=1 PSE_SIDE_PRIORITY=low"; do
  echo "== bench $v"; env $v $B 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['mf_evals_per_s'], d['mf_evals_per_s_moving'])"
done
for pr in default low; do
  echo "== replicated team solo, PSE_SIDE_PRIORITY=$pr"
  PSE_SIDE_PRIORITY=$pr python tools/perf_team.py --ranks 8 --steps 5 --solo 3 2>&1 | grep -E "solo|team of"
done
echo "== local"; python tools/perf_team.py --local --ranks 8 --steps 5 --solo 3 2>&1 | grep -E "solo|local team"
