#!/bin/bash
# SQ-level PMC passes for the bits GEMM (stall attribution).  usage: pmc_sq.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcsq; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 6 --warmup 2 --cpu-rows 0 --alt-operands none $@"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_CYCLES SQ_LEVEL_WAVES --kernel-trace --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
python3 scripts/pmc_summary.py $OUT | grep xf_bits
