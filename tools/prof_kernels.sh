#!/bin/bash
# Per-kernel profile on the GPU box (developer tool): kernel-trace stats + three SQ/TA counter passes of tools/perf.py.
#   tools/prof_kernels.sh <out-name> [perf.py args...]
# Output under gpurun_out/<out-name>/ ; summaries printed.
export TMPDIR=/tmp
# the interpreter itself after `--` (a launcher that re-execs under the profiler is refused on the GPU box)
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
N=$1; shift
O=gpurun_out/$N; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $PYREAL tools/perf.py "$@" > $O/stats.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/stats/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:26]:
    print(f"{r['Name'].split('(')[0].replace('void ','')[:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
if [ -z "$PSE_PROF_NOPMC" ]; then
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/a -- $PYREAL tools/perf.py "$@" > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/b -- $PYREAL tools/perf.py "$@" > $O/b.log 2>&1
timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/c -- $PYREAL tools/perf.py "$@" > $O/c.log 2>&1
python3 tools/pmc_table.py $O/a $O/b $O/c --like ${PSE_PROF_LIKE:-pse::k_} > $O/pmc.txt
fi
