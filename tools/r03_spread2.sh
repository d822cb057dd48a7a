#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/spread2; rm -rf $O; mkdir -p $O
PSE_SPREAD_MFMA=0 timeout 900 python -m pytest tests/test_reference_kernels.py -m gpu -q 2>&1 | tail -5 > $O/pytest_old.txt
timeout 900 python -m pytest tests/test_reference_kernels.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" > $O/pytest_new.txt
export PSE_PROF_LIKE=k_spread
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/a -- python3 tools/perf.py --only-mf --steps 2 > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $O/b -- python3 tools/perf.py --only-mf --steps 2 > $O/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/c -- python3 tools/perf.py --only-mf --steps 2 > $O/c.log 2>&1
python3 tools/pmc_table.py $O/a $O/b $O/c --like k_spread > $O/pmc.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/perf.py --only-mf --steps 5 > $O/stats.log 2>&1
python3 - $O <<'PY' > $O/stats.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/stats/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'].split('(')[0].replace('void ','')[:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
cat $O/pytest_old.txt $O/pytest_new.txt $O/pmc.txt $O/stats.txt; tail -3 $O/b.log
