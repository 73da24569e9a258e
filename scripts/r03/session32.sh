#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s32
timeout -k 10 600 python -m pytest tests/test_masked_gpu.py tests/test_link_gpu.py tests/test_palm_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 500 python bench.py --steps 10 --warmup 3 --cpu-rows 0 --traffic 0 --sustained 0 > gpurun_out/s32/bench.json 2> gpurun_out/s32/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/s32/bench.json').read().strip().splitlines()[-1])
print(round(d['value'],1), {k:(round(v['iterations_per_s'],1)) for k,v in d['secondary']['widened_engines'].items()})
P
