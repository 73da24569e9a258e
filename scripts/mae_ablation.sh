#!/bin/bash
# Where the MAE pass's time goes: build libbmf_hip with one ingredient of mae_kernel removed at a time (wrong results, same
# loop structure) and time bmf_mae_sum with each.  Numbers quoted in the header of csrc/mae.hip and DESIGN.md section 5.
#   usage (on the GPU box, through gpurun):  bash scripts/mae_ablation.sh
set -e
cd "$(dirname "$0")/../pybmf_amd/csrc"
LIBS="libbmf_hip.so"
NR="-DBMF_EXP_MAE_NO_REDUCE"
for v in NO_REDUCE:"$NR" NO_MFMA:"-DBMF_EXP_MAE_NO_MFMA" \
         MFMA_ONLY:"$NR -DBMF_EXP_MAE_NO_DMA -DBMF_EXP_MAE_NO_X -DBMF_EXP_MAE_NO_LDS" \
         MFMA_LDS:"$NR -DBMF_EXP_MAE_NO_DMA -DBMF_EXP_MAE_NO_X" MFMA_VDMA:"$NR -DBMF_EXP_MAE_NO_LDS -DBMF_EXP_MAE_NO_X" \
         MFMA_XDMA:"$NR -DBMF_EXP_MAE_NO_LDS -DBMF_EXP_MAE_NO_DMA" SKELETON:"$NR -DBMF_EXP_MAE_NO_MFMA"; do
    n=${v%%:*}; f=${v#*:}
    make -j8 OUT=libbmf_exp_$n.so BUILD=build_exp_$n EXTRA="$f" > /dev/null
    LIBS="$LIBS libbmf_exp_$n.so"
done
cd ../..
for l in $LIBS; do BMF_LIB=$l python scripts/mae_bench.py 2>/dev/null; done
rm -rf pybmf_amd/csrc/build_exp_* pybmf_amd/csrc/libbmf_exp_*.so
