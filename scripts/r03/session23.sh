#!/bin/bash
# MAE pass: x through fp4 converts vs fp8 converts (A/B in one box) + tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s23; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_wide_gpu.py -x -q -k "mae or wide" > $OUT/test.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/test.log
for f in 0 1 0 1; do BMF_MAE_FP4=$f BMF_MAE_TILED=1 timeout -k 10 120 python scripts/mae_bench.py 2>&1 | tail -1 | sed "s/^/fp4=$f /"; done
for f in 0 1 0 1; do
  BMF_MAE_FP4=$f timeout -k 10 200 python bench.py --mae 1 --secondary 0 --cpu-rows 0 --traffic 0 --alt-operands none --sustained 0 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
print("loop fp4=$f", "value", round(d["value"],1), "ms", round(d["ms_per_step"],4), "MAE", d["final"].get("MAE") if "final" in d else None)
PY
done
