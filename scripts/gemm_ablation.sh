#!/bin/bash
# Where the bits GEMM's time goes: build libbmf_hip with one ingredient of the kernel removed at a time (wrong results, same
# loop structure) and A/B them against the real library in one box.  Numbers quoted in DESIGN.md section 5.1.
#   usage (on the GPU box, through gpurun):  bash scripts/gemm_ablation.sh
set -e
cd "$(dirname "$0")/../pybmf_amd/csrc"
LIBS="libbmf_hip.so"
for v in NOVALU:"-DBMF_EXP_NOVALU" NOLDS:"-DBMF_EXP_NOLDS" NODMA:"-DBMF_EXP_NODMA" NOBAR:"-DBMF_EXP_NOBAR" WRAP:"-DBMF_EXP_PANEL_WRAP" \
         ALL:"-DBMF_EXP_NOVALU -DBMF_EXP_NOLDS -DBMF_EXP_NODMA -DBMF_EXP_NOBAR"; do
    n=${v%%:*}; f=${v#*:}
    make -j8 OUT=libbmf_exp_$n.so BUILD=build_exp_$n EXTRA="$f" > /dev/null
    LIBS="$LIBS libbmf_exp_$n.so"
done
cd ../..
# BMF_BENCH_TOL=-1: garbage factors must not trip the early stop; BMF_NO_CHECK=1: skip the result cross-checks
BMF_BENCH_TOL=-1 BMF_NO_CHECK=1 bash scripts/ab_bench.sh $LIBS
rm -rf pybmf_amd/csrc/build_exp_* pybmf_amd/csrc/libbmf_exp_*.so
