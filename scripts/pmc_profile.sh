#!/bin/bash
# PMC passes for the bench (run on the GPU box through gpurun).  Counters are collected in their own runs
# (--pmc with --kernel-trace only), one TCC group per pass (FETCH_SIZE and WRITE_SIZE do not fit together).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
mkdir -p $OUT
ARGS="bench.py --steps 6 --warmup 2 --cpu-rows 0 --alt-operands none"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- python3 $ARGS > $OUT/mfma.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
find $OUT -name "*.csv" | head -20
