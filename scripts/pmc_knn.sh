#!/bin/bash
# PMC passes for the kNN kernel (separate passes: SQ has 8 slots, FETCH_SIZE/WRITE_SIZE cannot share one)
export TMPDIR=/tmp
OUT=gpurun_out/pmc_knn_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/p1 --output-format csv -- python3 scripts/dev_knn_one.py $2 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $OUT/p2 --output-format csv -- python3 scripts/dev_knn_one.py $2 > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE -d $OUT/p3 --output-format csv -- python3 scripts/dev_knn_one.py $2 > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p4 --output-format csv -- python3 scripts/dev_knn_one.py $2 > $OUT/p4.log 2>&1
find $OUT -name "*counter_collection.csv" | head
