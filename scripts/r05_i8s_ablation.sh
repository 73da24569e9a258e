#!/bin/bash
# Round 5: where the sparse bits GEMM's time goes -- flavours of xf_bits_i8s.hip with one ingredient removed (wrong results on
# purpose), timed in one box.  Build here (no GPU needed): r05_i8s_ablation.sh build ; run on the box: r05_i8s_ablation.sh run
FLAVOURS="${FLAVOURS:-nodma noxdma nobar nolds novalu nomfma nodma_noxdma nodma_noxdma_nobar}"
if [ "$1" = build ]; then
  for f in $FLAVOURS; do
    flags=$(echo $f | tr '_' '\n' | sed 's/^/-DBMF_EXP_/' | tr 'a-z' 'A-Z' | tr '\n' ' ')
    bash scripts/build_flavour.sh s_$f "$flags" xf_bits_i8s.hip
  done
else
  mkdir -p gpurun_out/r05
  for f in base $FLAVOURS; do
    lib=libbmf_s_$f.so; [ "$f" = base ] && lib=libbmf_hip.so
    MODE=time FORM=${FORM:-0} BMF_LIB=$lib timeout -k 10 120 python scripts/r05_i8s_microbench.py 30 2>/dev/null | grep "sparse kernel"
  done
fi
