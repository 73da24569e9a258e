#!/bin/bash
# Round 5: what bounds the fused update epilogue -- flavours with one ingredient removed or replaced (the fp64 division, the digit
# extraction), timed alone at the U and the V shape in one box.  Build: the four build_flavour.sh lines of HISTORY.md; run on the GPU box.
for f in base epi_nodiv epi_fastdiv epi_nodigits epi_nodiv_nodigits; do
  lib=libbmf_$f.so; [ "$f" = base ] && lib=libbmf_hip.so
  for rows in 100352 20480; do
    echo -n "$f rows $rows: "; BMF_LIB=$lib timeout -k 10 120 python scripts/r03/epi_bench.py $rows 64 2 2>/dev/null | tail -1
  done
done
