export TMPDIR=/tmp
O=gpurun_out/prof_q; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/scratch/perf2.py > $O/log.txt 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_q/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print(f"{r['Name'].split('(')[0][:62]:62s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
