export TMPDIR=/tmp
python bench.py > gpurun_out/bench_v3.json 2> gpurun_out/bench_v3.err; tail -c 1800 gpurun_out/bench_v3.json
PSE_NO_OVERLAP=1 python bench.py --no-cpu > gpurun_out/bench_v3_noverlap.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/prof_v3.log 2>&1
PSE_NO_OVERLAP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc4_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc4_fetch.log 2>&1
PSE_NO_OVERLAP=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc4_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc4_write.log 2>&1
