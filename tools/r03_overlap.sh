#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/overlap; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_reference_kernels.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" > $O/pytest.txt
for cfg in "base:" "m55:PSE_SIDE_CUMASK=55555555" "m33:PSE_SIDE_CUMASK=33333333" "m0f:PSE_SIDE_CUMASK=0f0f0f0f" "m77:PSE_SIDE_CUMASK=77777777" "m11:PSE_SIDE_CUMASK=11111111" "phi:PSE_SIDE_PRIO=1" "plo:PSE_SIDE_PRIO=-1" "one:PSE_OVERLAP=-1"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e timeout 300 python bench.py --no-cpu --no-ref-grid --steps 30 --warmup 5 > $O/b_$n.json 2> $O/b_$n.err
  python3 -c "import json,sys; d=json.load(open('$O/b_$n.json')); print('$n', round(d['ms_per_step'],4), d['ms_per_step_percentiles']['p50'], round(d['mf_evals_per_s'],1))"
done
cat $O/pytest.txt
