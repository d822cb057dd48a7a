# A/B runs of one switch (developer scratch: edit freely)
B="python bench.py --steps 30 --warmup 10 --no-cpu --no-ref-grid --no-traffic"
for v in "PSE_LIST_SORT=0" "PSE_LIST_SORT=1" "PSE_LIST_SORT=0" "PSE_LIST_SORT=1"; do
  echo "== bench $v"; env $v $B 2>&1 | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['mf_evals_per_s'], d['mf_evals_per_s_moving'], d['phases_ms_per_step']['matvec'], d['phases_ms_per_step']['real'], d['phases_ms_per_step']['sort'])"
done
