#!/bin/bash
# Round 2 PMC passes for the int8 bits GEMM (run on the GPU box through gpurun).  Counters in their own runs (--pmc with
# --kernel-trace only), one TCC group per pass.  usage: pmc_i8.sh OUTDIR [bench args]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$1; shift
mkdir -p $OUT
ARGS="bench.py --steps 6 --warmup 2 --cpu-rows 0 --alt-operands none --secondary 0 --traffic 0 $@"   # (--traffic 0: no profiler children inside a profiled run)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -- python3 $ARGS > $OUT/tcc.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- python3 $ARGS > $OUT/mfma.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
python3 scripts/pmc_summary.py $OUT > $OUT/summary.md
cat $OUT/summary.md
