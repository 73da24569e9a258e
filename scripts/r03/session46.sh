#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_masked_gpu.py tests/test_properties_gpu.py tests/test_penalty_gpu.py tests/test_edge_cases_gpu.py tests/test_link_gpu.py tests/test_palm_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 300 python scripts/r03/prediction_bench.py 2>&1 | tail -2
