set -x
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_v0.json 2> gpurun_out/bench_v0.err; tail -3 gpurun_out/bench_v0.err; cat gpurun_out/bench_v0.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v0 -- python3 bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/prof_v0.log 2>&1
tail -3 gpurun_out/prof_v0.log
find gpurun_out/prof_v0 -name "*stats*" | head
