#!/bin/bash
# PMC passes over the int8 bits-GEMM microbenchmark (separate --pmc runs with --kernel-trace only): r04_pmc_i8w.sh OUTNAME [ENV=VAL ...]
# e.g.  r04_pmc_i8w.sh narrow BMF_I8_WIDE=0 ;  r04_pmc_i8w.sh wide BMF_I8_WIDE=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04_pmc/$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $OUT
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
P3="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -- python3 scripts/gemm_i8_microbench.py 12 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
python3 scripts/pmc_summary.py $OUT | grep "xf_bits_i8" > $OUT/summary.md
cat $OUT/summary.md
