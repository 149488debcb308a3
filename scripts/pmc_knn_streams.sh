#!/bin/bash
# dev helper: SQ / LDS counters of the two fp16 kNN streams (three --pmc passes).  usage: scripts/pmc_knn_streams.sh TAG
export TMPDIR=/tmp
OUT=gpurun_out/pmc_knn_$1; shift
rm -rf $OUT; mkdir -p $OUT
A="--steps 3 --warmup 2 --no-cpu-baseline --no-train-leg --no-model-legs --no-graph-leg"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/p1 --output-format csv -- python3 bench.py $A > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY -d $OUT/p2 --output-format csv -- python3 bench.py $A > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -d $OUT/p3 --output-format csv -- python3 bench.py $A > $OUT/p3.log 2>&1
python3 scripts/pmc_summary.py $OUT knn_bound16s knn_filter16
