#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_models_gpu.py tests/test_properties_gpu.py tests/test_masked_gpu.py -x -q -m gpu -k "thresh or Thresh or line_search or threshold" 2>&1 | tail -2
BMF_THRESH_POLL=0 timeout -k 10 200 python scripts/c5_bench.py 2>&1 | tail -1
timeout -k 10 200 python scripts/c5_bench.py 2>&1 | tail -1
timeout -k 10 200 python scripts/c5_bench.py 2>&1 | tail -1
