export TMPDIR=/tmp
python bench.py > gpurun_out/bench_v1.json 2> gpurun_out/bench_v1.err; tail -1 gpurun_out/bench_v1.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/prof_v1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/pmc_write.log 2>&1
ls gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/ | head
