#!/bin/bash
# Round check on the GPU box: GPU test suite, one-rank sharded bench under torch.distributed.run, default bench,
# kernel-trace profile, HBM traffic counters (separate passes).  Outputs under gpurun_out/check/.
export TMPDIR=/tmp
# the interpreter itself after `--` (a launcher that re-execs under the profiler is refused on the GPU box); profiled runs time the
# 256^3 engine alone: no second 360^3 engine, no nested counter passes (their kernels would be mixed into the averages)
PYREAL=$(python3 -c 'import os,sys; print(os.path.realpath(sys.executable))')
PROF="--no-cpu --no-ref-grid --no-traffic"
O=gpurun_out/check; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" > $O/pytest.txt
PSE_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > $O/bench_sharded1.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
PSE_OVERLAP=1 timeout 600 python bench.py --no-cpu > $O/bench_overlap.json 2> $O/bench_overlap.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $PYREAL bench.py --steps 10 --warmup 3 $PROF > $O/prof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $PYREAL bench.py --steps 3 --warmup 1 $PROF > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $PYREAL bench.py --steps 3 --warmup 1 $PROF > $O/pmc_write.log 2>&1
cat $O/pytest.txt; tail -1 $O/bench_sharded1.txt | cut -c1-400; tail -c 700 $O/bench.json; echo; tail -c 400 $O/bench_overlap.json
