#!/bin/bash
# Config #2 GEMM kernels, library flavours x reduction splits in one box: r04_f32_flavours.sh "base f32r3 ..." [SPLITS] [SPLITS_T]
for f in $1; do
  lib=libbmf_$f.so; [ "$f" = base ] && lib=libbmf_hip.so
  echo "== $f"; SPLITS=${2:-2,3,4,6} SPLITS_T=${3:-6,9,12,13} BMF_LIB=$lib timeout -k 10 200 python scripts/xf_f32_microbench.py 2>&1 | grep "us,"
done
