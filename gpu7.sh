export TMPDIR=/tmp
python bench.py > gpurun_out/bench_v2.json 2> gpurun_out/bench_v2.err; tail -c 2500 gpurun_out/bench_v2.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v2 -- python3 bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/prof_v2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc2_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc2_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc2_write.log 2>&1
ls gpurun_out/prof_v2/*/ gpurun_out/pmc2_fetch/*/ gpurun_out/pmc2_write/*/ | grep -E "stats|counter"
