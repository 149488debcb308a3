#!/bin/bash
# Evidence behind profiles/<tag>_*: rocprofv3 kernel statistics of the three workloads bench.py knows, then the PMC
# passes of the headline workload -- each counter group in its OWN pass with --kernel-trace only (never combined with
# --sys-trace & co.; FETCH_SIZE and WRITE_SIZE cannot share a pass).   usage: scripts/profile_round.sh r2_10
export TMPDIR=/tmp
TAG=$1
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-train-leg --no-model-legs --no-graph-leg"
rocprofv3 --kernel-trace --stats -d $OUT/block --output-format csv -- python3 bench.py --steps 20 --warmup 3 $B > $OUT/block.log 2>&1
echo block done
rocprofv3 --kernel-trace --stats -d $OUT/two --output-format csv -- python3 bench.py --steps 20 --warmup 3 --stages 2 $B > $OUT/two.log 2>&1
echo two-stage done
rocprofv3 --kernel-trace --stats -d $OUT/train --output-format csv -- python3 bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline > $OUT/train.log 2>&1
echo train done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- python3 bench.py --steps 4 --warmup 2 $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write --output-format csv -- python3 bench.py --steps 4 --warmup 2 $B > $OUT/write.log 2>&1
echo traffic done
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d $OUT/sq --output-format csv -- python3 bench.py --steps 4 --warmup 2 $B > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d $OUT/sq2 --output-format csv -- python3 bench.py --steps 4 --warmup 2 --stages 2 $B > $OUT/sq2.log 2>&1
echo sq done
python3 scripts/profile_summaries.py $OUT $TAG
