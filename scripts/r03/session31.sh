#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s31
timeout -k 10 500 python -m pytest tests/test_palm_gpu.py -x -q -m gpu 2>&1 | tail -5
BMF_PALM_FUSED=0 timeout -k 10 200 python scripts/palm_bench.py
timeout -k 10 200 python scripts/palm_bench.py
