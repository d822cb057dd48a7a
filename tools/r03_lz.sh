#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/lz; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_nlist.py tests/test_gpu_slabs.py tests/test_gpu_host.py -m gpu -x -q 2>&1 | tail -6 > $O/pytest.txt
for r in 1 2; do timeout 300 python bench.py --no-cpu --no-ref-grid --steps 30 --warmup 5 > $O/b$r.json 2> $O/b$r.err; python3 -c "import json; d=json.load(open('$O/b$r.json')); print(round(d['ms_per_step'],4), d['ms_per_step_percentiles']['p50'], d['phases_ms_per_step'])"; done
cat $O/pytest.txt
