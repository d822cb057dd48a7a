#!/bin/bash
# Round check on the GPU box: GPU test suite, one-rank sharded bench under torch.distributed.run, default bench,
# kernel-trace profile.  Outputs under gpurun_out/check/.
export TMPDIR=/tmp
O=gpurun_out/check; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" > $O/pytest.txt
PSE_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $O/bench_sharded1.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu > $O/prof.log 2>&1
tail -3 $O/pytest.txt; tail -2 $O/bench_sharded1.txt; tail -c 600 $O/bench.json
