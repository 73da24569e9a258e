#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s5; mkdir -p $OUT
export TMPDIR=/tmp
python scripts/r03/epi_bench.py 2>&1 | tail -1 | tee $OUT/epi.txt
python scripts/r03/epi_bench.py 20480 64 1 2>&1 | tail -1 | tee -a $OUT/epi.txt
python -m pytest tests/test_kernels_gpu.py tests/test_edge_cases_gpu.py tests/test_penalty_gpu.py "tests/test_sharded_gpu.py::test_sharded_engine_matches_single" -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
grep -E "passed|failed|^FAILED|differs|Error" $OUT/pytest.log | tail -15
for cfg in "0 0.5" "1 0.5" "0 0.63" "1 0.63"; do
  set -- $cfg
  echo "== bench fused $1 share $2"
  BMF_I8_FUSED_PLANES=$1 BMF_I8_OLD_SHARE=$2 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/bench_$1_$2.err | tail -1 > $OUT/bench_$1_$2.json
  python -c "
import json; d=json.load(open('$OUT/bench_$1_$2.json')); print('%.4f ms/step %.1f it/s gemm %.1f us cold %.1f' % (d['ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms'], d['cold_start']['value']))"
done | tee $OUT/bench_ab.txt
