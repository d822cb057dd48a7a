export TMPDIR=/tmp
O=gpurun_out/pmc_mv; rm -rf $O; mkdir -p $O
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $O/a -- python3 tools/scratch/perf2.py > $O/a.log 2>&1
rocprofv3 --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $O/b -- python3 tools/scratch/perf2.py > $O/b.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum --kernel-trace --output-format csv -d $O/c -- python3 tools/scratch/perf2.py > $O/c.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/d -- python3 tools/scratch/perf2.py > $O/d.log 2>&1
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --kernel-trace --output-format csv -d $O/e -- python3 tools/scratch/perf2.py > $O/e.log 2>&1
tail -2 $O/*.log | cut -c1-200
find $O -name "*counter_collection.csv" | xargs ls -la
