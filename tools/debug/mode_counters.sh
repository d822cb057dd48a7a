#!/bin/bash
# The two speeds of the 512^3 x pass: counter passes over tools/debug/mode_probe.py (four engines alive in one
# process: their grids land on different physical pages).  Output gpurun_out/<tag>/mode_*.txt
export TMPDIR=/tmp
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
T=${1:-mode}; O=gpurun_out/$T; mkdir -p $O
timeout 600 python3 tools/debug/mode_probe.py 512 4 > $O/mode_times.txt 2>&1
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc$i -- $PYREAL tools/debug/mode_probe.py 512 4 > $O/pmc$i.log 2>&1
  { echo "# --pmc $set"; python3 tools/debug/mode_counters.py $O/pmc$i; } > $O/mode_counters_$i.txt 2>&1
done
cat $O/mode_times.txt; head -30 $O/mode_counters_1.txt
