export TMPDIR=/tmp
PSE_FORCE_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu 2>&1 | tail -1
