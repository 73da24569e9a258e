#!/bin/bash
# link pass at two waves per SIMD: tests + rates at the headline shape
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s25; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_link_gpu.py -x -q > $OUT/test.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/test.log
timeout -k 10 300 python scripts/link_bench.py 2>&1 | grep "update"
