export TMPDIR=/tmp
rocprofv3 --list-avail > gpurun_out/avail.txt 2>&1
grep -c . gpurun_out/avail.txt
