#!/bin/bash
# MAE pass: does the grid's last partial round matter?  384 row blocks x 8 column groups = 4 x 768 resident workgroups exactly
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for m in 100000 98304 100000 98304; do
  M=$m BMF_MAE_TILED=1 timeout -k 10 120 python scripts/mae_bench.py 2>&1 | tail -1
done
for g in 6 12; do M=98304 BMF_MAE_GROUPS=$g BMF_MAE_TILED=1 timeout -k 10 120 python scripts/mae_bench.py 2>&1 | tail -1; done
