#!/bin/bash
# A/B of kernel flavours in one box: ab_i8.sh "flavour1 flavour2 ..." [launches]
mkdir -p gpurun_out/r02
for f in $1; do
  lib=libbmf_$f.so; [ "$f" = base ] && lib=libbmf_hip.so
  BMF_LIB=$lib timeout -k 10 120 python scripts/gemm_i8_microbench.py ${2:-40} 2>/dev/null | tail -1
done
