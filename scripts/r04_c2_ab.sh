#!/bin/bash
# Config #2 C-side loop: fused update (one launch per side) against the epilogue + re-order + Gram launches, and the split rule.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  echo -n "fused update: "; timeout -k 10 120 python scripts/c2_loop.py 2>&1 | tail -1
  echo -n "separate launches: "; BMF_C2_FUSED_UPDATE=0 timeout -k 10 120 python scripts/c2_loop.py 2>&1 | tail -1
  echo -n "fused, X^T U splits 13: "; BMF_F32_SPLITS_XTU=13 timeout -k 10 120 python scripts/c2_loop.py 2>&1 | tail -1
  echo -n "fused, X^T U splits 9: "; BMF_F32_SPLITS_XTU=9 timeout -k 10 120 python scripts/c2_loop.py 2>&1 | tail -1
done
