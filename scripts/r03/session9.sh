#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s9; mkdir -p $OUT
export TMPDIR=/tmp
for lead in 0 1; do
  echo "== lead $lead"
  BMF_I8_LEAD=$lead timeout -k 10 200 python scripts/gemm_i8_microbench.py 40 2>&1 | tail -1
  BMF_I8_LEAD=$lead timeout -k 10 400 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --secondary 0 --sustained 0 2>$OUT/bench_$lead.err | tail -1 > $OUT/bench_$lead.json
  python -c "
import json; d=json.load(open('$OUT/bench_$lead.json')); r=d['roofline']; print('%.4f ms/step %.1f it/s gemm %.1f us traffic %.3f GB (%.2fx)' % (d['ms_per_step'], d['value'], 1e3*r['avg_launch_ms'], (r['traffic'] or 0)/1e9, r['traffic_over_algorithmic'] or 0))"
done | tee $OUT/lead.txt
python -m pytest tests/test_kernels_gpu.py tests/test_penalty_gpu.py -m gpu -q -x 2>&1 | tail -2
