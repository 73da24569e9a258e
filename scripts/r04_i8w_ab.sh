#!/bin/bash
# A/B of the two int8 bits-GEMM kernels in one box: correctness of the 64-column kernel first, then the microbenchmark with
# BMF_I8_WIDE=0 (32-column kernel, two 4-wave workgroups per CU) and =1 (64-column kernel, one 8-wave workgroup per CU), interleaved.
mkdir -p gpurun_out/r04
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "i8" 2>&1 | tail -5 || exit 1
for rep in 1 2; do
  for w in 0 1; do
    echo -n "WIDE=$w "; BMF_I8_WIDE=$w timeout -k 10 120 python scripts/gemm_i8_microbench.py ${1:-40} 2>/dev/null | tail -1
  done
done
