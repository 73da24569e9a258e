#!/bin/bash
# config #2 after merging the three small launches in front of the fused pass: tests + rate
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s19; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_properties_gpu.py tests/test_kernels_gpu.py -x -q -k "wnmf or WNMF or real or config2 or tiled" > $OUT/test2.log 2>&1; echo "model tests rc=$?"; tail -3 $OUT/test2.log
for f in 1 1; do
  BMF_C2_FUSED_RESID=$f timeout -k 10 200 python - <<'PY'
import os, sys, json
sys.path.insert(0, os.getcwd())
import bench
r = bench.secondary_c2()
print("fused", os.environ["BMF_C2_FUSED_RESID"], "it/s %.1f ms %.4f err %.6f" % (r["iterations_per_s"], r["ms_per_iteration"], r["error"]))
PY
done
