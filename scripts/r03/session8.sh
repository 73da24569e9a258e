#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s8; mkdir -p $OUT
export TMPDIR=/tmp
./scripts/probes/i8_shape_probe 2>&1 | tee $OUT/i8_shape_probe.txt
python -m pytest tests/test_kernels_gpu.py tests/test_penalty_gpu.py tests/test_edge_cases_gpu.py tests/test_sharded_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
grep -E "passed|failed|^FAILED" $OUT/pytest.log | tail -5
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/bench.err | tail -1 > $OUT/bench.json
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('%.4f ms/step %.1f it/s gemm %.1f us cold %.1f' % (d['ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms'], d['cold_start']['value']))"
for mm in 12500; do
  for mode in 0 1; do
    BMF_FORCE_SHARDED=$mode timeout -k 10 300 python bench.py --m $mm --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/bench_m${mm}_s${mode}.err | tail -1 > $OUT/bench_m${mm}_s${mode}.json
    python -c "
import json; d=json.load(open('$OUT/bench_m${mm}_s${mode}.json')); print('m=$mm sharded=$mode: %.4f ms/step %.1f it/s gemm %.1f us' % (d['ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']))"
  done
done
