#!/bin/bash
# round 3, first GPU call: fp64 MFMA rate, FETCH_SIZE calibration for the list stream, ablation of the pair-list mat-vec
export TMPDIR=/tmp
O=gpurun_out/probe1; rm -rf $O; mkdir -p $O
timeout 120 tools/microbench/mfma_f64 > $O/mfma.txt 2>&1
timeout 120 tools/microbench/nt_fetch > $O/nt_fetch.txt 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/ntpmc -- tools/microbench/nt_fetch > $O/ntpmc.log 2>&1
python3 - $O <<'PY' > $O/nt_pmc.txt 2>&1
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/ntpmc/**/*counter_collection.csv',recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    d[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k,v in d.items(): print(k, [round(x/1024,1) for x in v], 'MB (FETCH_SIZE is in KB)')
PY
timeout 300 python3 tools/perf.py --steps 5 > $O/perf_base.txt 2>&1
for v in 1 2 3; do timeout 300 python3 tools/run_with_lib.py tools/ablate/libpse_abl$v.so tools/perf.py --steps 3 > $O/perf_abl$v.txt 2>&1; done
cat $O/mfma.txt $O/nt_fetch.txt $O/nt_pmc.txt; grep -h "Brownian phases" $O/perf_*.txt
