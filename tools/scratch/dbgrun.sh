export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror|assert" | head
for d in 0 4 12; do PSE_DBG=$d python tools/scratch/perf2.py 2>&1 | tail -1 | cut -c100-400; done
PSE_NO_SEG=1 python tools/scratch/perf2.py 2>&1 | tail -1 | cut -c100-400
