#!/bin/bash
# the developer switches of spread and gather against their automatic choices at the metric point (one stream): step time and phases
run() { env "$@" PSE_OVERLAP=-1 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-ref-grid --no-cfg4 --no-traffic 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['phases_ms_per_step']; print('$*', round(d['ms_per_step'],4), {k:p[k] for k in ('spread','gather','records')})"; }
run X=0
run PSE_SPREAD_TZ=8
run PSE_SPREAD_TZ=16
run PSE_SPREAD_NW=1
run PSE_SPREAD_NW=2
run PSE_SPREAD_NW=4
run PSE_GATHER_BZ=1
run PSE_GATHER_BZ=2
run X=0
