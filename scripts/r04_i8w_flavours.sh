#!/bin/bash
# microbenchmark of library flavours in one box: r04_i8w_flavours.sh "name1[:WIDE] name2 ..." [launches]   (base = libbmf_hip.so; :WIDE = BMF_I8_WIDE variant)
for rep in 1 2; do
for fw in $1; do
  f=${fw%%:*}; w=${fw##*:}; [ "$w" = "$fw" ] && w=1
  lib=libbmf_$f.so; [ "$f" = base ] && lib=libbmf_hip.so
  echo -n "$f wide=$w: "; BMF_I8_WIDE=$w BMF_LIB=$lib timeout -k 10 120 python scripts/gemm_i8_microbench.py ${2:-40} 2>/dev/null | tail -1 | sed 's/.*libbmf[^ ]* //'
done; done
