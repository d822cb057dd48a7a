export TMPDIR=/tmp
O=gpurun_out/pmc_sp; rm -rf $O; mkdir -p $O
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/a -- python3 tools/scratch/perf2.py > $O/a.log 2>&1
timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/b -- python3 tools/scratch/perf2.py > $O/b.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/c -- python3 tools/scratch/perf2.py > $O/c.log 2>&1
ls $O/*/*/ | head
