#!/bin/bash
# the two-block engine for 64 < k <= 128 against the oracle
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s21; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_wide_gpu.py tests/test_abi.py -x -q > $OUT/test.log 2>&1; echo "tests rc=$?"; tail -25 $OUT/test.log
