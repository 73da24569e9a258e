#!/bin/bash
# Where mae32_kernel's time goes: flavours with one ingredient removed (wrong results, same loop), A/B in one box.
set -e
cd "$GRAFT_REPO_ROOT"
for l in hip m32_NO_XDMA m32_NO_VDMA; do
  BMF_LIB=libbmf_$l.so timeout -k 10 100 python scripts/mae_bench.py 2>/dev/null | tail -1
done
for g in 8 13 15 17 19 23 27; do
  echo "groups $g: $(BMF_MAE_GROUPS=$g timeout -k 10 100 python scripts/mae_bench.py 2>/dev/null | tail -1 | cut -c1-90)"
done
