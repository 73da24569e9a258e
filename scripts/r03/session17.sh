#!/bin/bash
# PALM engine: tests + ELBMF loop rate at the headline shape after the sym_norms rewrite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s17; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_palm_gpu.py tests/test_kernels_gpu.py -x -q -k "palm or norm or elbmf or primp" > $OUT/test.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/test.log
timeout -k 10 300 python scripts/palm_bench.py 2>&1 | tail -1
