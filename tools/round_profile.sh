#!/bin/bash
# Round evidence on the GPU box: bench line, kernel-trace profile of the same command, HBM traffic counters (separate passes, as
# MI355X_MICROARCH.md prescribes), SQ/TA counters.  tools/round_profile.sh <tag> <commit>  -> gpurun_out/<tag>/ ; the summaries
# (kernel_stats.csv, pmc_traffic.json with the commit stamped in, sq_counters.txt) are written there too: copy them into profiles/.
export TMPDIR=/tmp
# (--no-cfg4 --no-async in the profiled commands: the config-4 block launches kernels of the same names, and the gated extra iterations of
# queue-only steps are launches that leave at once: both would pollute per-kernel averages)
# the interpreter itself after `--` (a launcher that re-execs under the profiler is refused on the GPU box)
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
T=${1:-r03}
C=${2:-unrecorded}
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
PSE_OVERLAP=0 timeout 600 python bench.py --no-cpu --no-ref-grid --no-cfg4 --no-traffic > $O/bench_one_stream_steps.json 2> $O/bench_one_stream_steps.err   # Brownian steps on ONE stream (the default of rounds 1-5), for comparison
export PSE_OVERLAP=-1   # profiles: every kernel alone on one stream (0 would still fork the far-field chain of the deterministic M.F evaluations
                        # of the bench: their overlapped launches, 300-690 us for a 143 us y pass, were in the per-kernel averages of rounds 1-5)
CMD="$PYREAL bench.py --steps 10 --warmup 3 --no-cpu --no-ref-grid --no-cfg4 --no-async --no-traffic"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $CMD > $O/prof.log 2>&1
CMD3="$PYREAL bench.py --steps 3 --warmup 1 --no-cpu --no-ref-grid --no-cfg4 --no-async --no-traffic"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $CMD3 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $CMD3 > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/sq_a -- $CMD3 > $O/sq_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC --kernel-trace --output-format csv -d $O/sq_b -- $CMD3 > $O/sq_b.log 2>&1
timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_c -- $CMD3 > $O/sq_c.log 2>&1
python3 tools/summarize_prof.py stats $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv "PSE_OVERLAP=-1 $CMD"
python3 tools/summarize_prof.py pmc $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json "$C" "PSE_OVERLAP=-1 $CMD3"
python3 tools/pmc_table.py $O/sq_a $O/sq_b $O/sq_c --like pse::k_ > $O/sq_counters.txt
tail -c 600 $O/bench.json; echo; head -12 $O/kernel_stats.csv
