#!/bin/bash
# bench loop with the MAE pass after the launch merge + tests that exercise the MAE column
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s13; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_penalty_gpu.py tests/test_models_gpu.py -x -q > $OUT/test.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/test.log
for sh in 32 32; do
  BMF_MAE_SHAPE=$sh timeout -k 10 200 python bench.py --mae 1 --secondary 0 --cpu-rows 0 --traffic 0 --alt-operands none --sustained 0 > $OUT/b_$sh.json 2> $OUT/b_$sh.err
  python - <<PY
import json
d=json.loads(open("$OUT/b_$sh.json").read().strip().splitlines()[-1])
print("shape $sh", "value", round(d["value"],1), "ms", round(d["ms_per_step"],4), d["roofline"].get("avg_launch_ms"))
PY
done
