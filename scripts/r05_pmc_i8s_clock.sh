#!/bin/bash
# Round 5: the clock the chip holds under the sparse bits GEMM's forms and ablation flavours (GRBM_GUI_ACTIVE / kernel duration), one
# --pmc pass per flavour (kernel trace only).  Flavours: scripts/r05_i8s_ablation.sh build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05/pmc_i8s_clock
mkdir -p $OUT
export MODE=time WITH_DENSE=1
for spec in "0:libbmf_hip.so" "2:libbmf_hip.so" "0:libbmf_s_nodma_noxdma_nobar_nolds_novalu.so" "0:libbmf_s_nodma_noxdma_nobar.so" "0:libbmf_s_nomfma.so"; do
  F=${spec%%:*}; LIBN=${spec##*:}; tag="form${F}_${LIBN%.so}"
  FORM=$F BMF_LIB=$LIBN timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/$tag -- python3 scripts/r05_i8s_microbench.py 10 > $OUT/$tag.log 2>&1 || { echo "$tag failed"; tail -5 $OUT/$tag.log; exit 1; }
  echo "== $tag"
  python3 scripts/pmc_summary.py $OUT/$tag | grep -E "xf_bits_i8"
done
