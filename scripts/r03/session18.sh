#!/bin/bash
# bench with the widened-engine leg
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s18; mkdir -p $OUT
timeout -k 10 500 python bench.py --cpu-rows 0 --traffic 0 --sustained 0 > $OUT/b.json 2> $OUT/b.err; echo "bench rc=$?"; tail -3 $OUT/b.err
python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "with_mae", round(d["with_mae"]["value"],1))
print(json.dumps({k: (round(v["iterations_per_s"], 1) if "iterations_per_s" in v else v) for k, v in d["secondary"]["widened_engines"].items()}))
PY
