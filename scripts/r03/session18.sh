#!/bin/bash
# bench with the widened-engine leg + palm tests after the scalars kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s18; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_palm_gpu.py tests/test_abi.py -x -q > $OUT/test.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/test.log
timeout -k 10 500 python bench.py --cpu-rows 0 --traffic 0 --sustained 0 > $OUT/b.json 2> $OUT/b.err; echo "bench rc=$?"; tail -3 $OUT/b.err
python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "with_mae", round(d["with_mae"]["value"],1))
print(json.dumps(d["secondary"]["widened_engines"], indent=1))
print({k: (v.get("iterations_per_s") if isinstance(v, dict) else v) for k, v in d["secondary"].items() if k != "widened_engines"})
PY
