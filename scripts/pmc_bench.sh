#!/bin/bash
# SQ counter passes over bench.py (separate passes; never combined with tracing domains other than --kernel-trace)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_bench_$1; shift
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/p1 --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/p2 --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $OUT/p3 --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > $OUT/p3.log 2>&1
true
