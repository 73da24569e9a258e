#!/bin/bash
# X^T U slab slots: tests + bench at full size and at the 8-GPU shard size
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s22; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_penalty_gpu.py tests/test_edge_cases_gpu.py tests/test_sharded_gpu.py tests/test_models_gpu.py -x -q > $OUT/test.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/test.log
for m in 100000 100000 12500; do
  timeout -k 10 200 python bench.py --m $m --secondary 0 --cpu-rows 0 --traffic 0 --alt-operands none --sustained 0 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
print("m $m", "value", round(d["value"],1), "ms", round(d["ms_per_step"],4))
PY
done
