#!/bin/bash
# Round evidence on the GPU box: bench line, kernel-trace profile of the same command, HBM traffic counters (separate passes, as
# MI355X_MICROARCH.md prescribes), SQ/TA counters.  tools/round_profile.sh <tag>  -> gpurun_out/<tag>/ ; summarise with
# tools/summarize_prof.py and tools/pmc_table.py into profiles/.
export TMPDIR=/tmp
T=${1:-r02}
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
PSE_OVERLAP=0 timeout 600 python bench.py --no-cpu > $O/bench_one_stream.json 2> $O/bench_one_stream.err
export PSE_OVERLAP=0    # profiles: every kernel alone on one stream
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu > $O/prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/sq_a -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/sq_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC --kernel-trace --output-format csv -d $O/sq_b -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/sq_b.log 2>&1
timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_c -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $O/sq_c.log 2>&1
tail -c 600 $O/bench.json; echo; ls $O
