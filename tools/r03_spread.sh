#!/bin/bash
# spread A/B on the GPU box: parity tests that exercise the far field, then phase times with the matrix-pipe spread on and off
export TMPDIR=/tmp
O=gpurun_out/spread; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_reference_kernels.py tests/test_gpu_parity.py -m gpu -x -q -k "spread or wave or mobility or kat or fused or device" 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python3 tools/perf.py --steps 5 --only-mf > $O/perf_mfma.txt 2>&1
PSE_SPREAD_MFMA=0 timeout 300 python3 tools/perf.py --steps 5 --only-mf > $O/perf_old.txt 2>&1
timeout 300 python3 tools/perf.py --steps 5 --only-mf --xy 0.3 > $O/perf_mfma_shear.txt 2>&1
cat $O/pytest.txt; grep -h "M.F phases" $O/perf_*.txt
