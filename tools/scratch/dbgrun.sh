export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror|assert" | head
python tools/scratch/perf2.py 2>&1 | tail -1 | cut -c1-300
