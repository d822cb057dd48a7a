export TMPDIR=/tmp
O=gpurun_out/pmc_final; rm -rf $O; mkdir -p $O
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/a -- python3 tools/scratch/perf2.py > $O/a.log 2>&1
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC --kernel-trace --output-format csv -d $O/b -- python3 tools/scratch/perf2.py > $O/b.log 2>&1
timeout 240 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum --kernel-trace --output-format csv -d $O/c -- python3 tools/scratch/perf2.py > $O/c.log 2>&1
ls $O/*/*/*counter_collection.csv
