#!/bin/bash
# Round 3, GPU session 4: digit planes emitted by the epilogue -- correctness (kernel test, trajectories, full-size parity), A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s4; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py tests/test_penalty_gpu.py tests/test_edge_cases_gpu.py tests/test_models_gpu.py tests/test_palm_gpu.py tests/test_sharded_gpu.py tests/test_c3_parity_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
grep -E "passed|failed|c3 parity|^FAILED|Error" $OUT/pytest.log | tail -15
for f in 0 1; do
  echo "== bench fused $f"
  BMF_I8_FUSED_PLANES=$f timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/bench_f$f.err | tail -1 > $OUT/bench_f$f.json
  python -c "
import json; d=json.load(open('$OUT/bench_f$f.json')); print('%.4f ms/step %.1f it/s gemm %.1f us' % (d['ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']), d['checks'])"
done | tee $OUT/bench_fused.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 --preheat 0 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/prof/*/*kernel_stats.csv | head -1); head -25 $f | cut -c1-150
