#!/bin/bash
# Config #2, X V pass: the LDS-ring kernel against the direct-to-register form (BMF_F32_DIRECT=1), same box, with a correctness check
for d in 0 1 0 1; do
  echo "== BMF_F32_DIRECT=$d"
  CHECK=1 SPLITS=${1:-2,3,4,6,8} SPLITS_T=6 BMF_F32_DIRECT=$d timeout -k 10 200 python scripts/xf_f32_microbench.py 2>&1 | grep -E "xf_f32_tiled |check|Error|error" || exit 1
done
