#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_palm_gpu.py tests/test_models_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 200 python scripts/palm_bench.py
