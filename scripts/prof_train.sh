#!/bin/bash
# rocprofv3 kernel statistics of the training step at a given global batch   usage: scripts/prof_train.sh <tag> <B>
export TMPDIR=/tmp
O=gpurun_out/prof_train_$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/run --output-format csv -- python3 bench.py --workload train --batch $2 --steps 6 --warmup 2 --no-cpu-baseline > $O/run.log 2>&1
cp $(ls $O/run/*/*kernel_stats.csv | head -1) gpurun_out/$1_kernel_stats.csv
python3 scripts/trace_gaps.py $O/run
