timeout 900 python -m pytest tests/test_gpu_slabs.py -m gpu -x -q 2>&1 | tail -30
