#!/bin/bash
# one PMC pass: MFMA busy + clock for the bits GEMM
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/mfma -- python3 bench.py --steps 6 --warmup 2 --cpu-rows 0 --alt-operands none "$@" > $OUT/mfma.log 2>&1
python3 scripts/pmc_summary.py $OUT | grep xf_bits
