#!/bin/bash
# Precision of a 19.8-bit quantisation (|q| <= 461 760 = what four balanced base-31 digits hold: the FP6 planes of the planned
# FP4 x FP6 bits GEMM), emulated exactly on the int8 x 3 path: same integer q, same exact accumulation.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s28
export BMF_LIB=libbmf_q19.so
timeout -k 10 400 python -m pytest tests/test_c3_parity_gpu.py -q -s -m gpu > gpurun_out/s28/c3.log 2>&1; echo "c3 rc=$?"; grep "c3 parity" gpurun_out/s28/c3.log
timeout -k 10 500 python -m pytest tests/test_penalty_gpu.py tests/test_models_gpu.py tests/test_edge_cases_gpu.py -q -m gpu > gpurun_out/s28/suite.log 2>&1; echo "suite rc=$?"; tail -15 gpurun_out/s28/suite.log
