set -x
python -c "import torch; print(torch.cuda.is_available(), torch.cuda.get_device_name(0))"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -40
