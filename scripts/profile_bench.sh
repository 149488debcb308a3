#!/bin/bash
# rocprofv3 runs behind profiles/: kernel stats, then HBM traffic counters in their own passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with --sys-trace etc.)
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/write.log 2>&1
grep -h '"metric"' $OUT/stats.log | cut -c1-160
