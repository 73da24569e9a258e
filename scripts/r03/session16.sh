#!/bin/bash
# fused finalize: tests + A/B in the bench loop at full size and at the 8-GPU shard size
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s16; mkdir -p $OUT
echo skip tests
for m in 100000 12500; do for f in 0 1 0 1; do
  BMF_FUSED_FINALIZE=$f timeout -k 10 200 python bench.py --m $m --secondary 0 --cpu-rows 0 --traffic 0 --alt-operands none --sustained 0 > $OUT/b.json 2> $OUT/b.err
  python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
print("m $m fused $f", "value", round(d["value"],1), "ms", round(d["ms_per_step"],4))
PY
done; done
