export TMPDIR=/tmp
python perf2.py 2>&1 | tail -1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_sq1 -- python3 perf2.py > gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -- python3 perf2.py > gpurun_out/pmc_sq2.log 2>&1
tail -2 gpurun_out/pmc_sq2.log
