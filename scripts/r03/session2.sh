#!/bin/bash
# Round 3, GPU session 2: the tests session 1 did not reach, the one-block exchange on the compute stream, per-workgroup stamps of the GEMM
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s2; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_palm_gpu.py tests/test_penalty_gpu.py tests/test_prediction_gpu.py tests/test_properties_gpu.py tests/test_sharded_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -25 $OUT/pytest.log
for mm in 12500 100000; do
  for mode in 0 1; do
    BMF_FORCE_SHARDED=$mode timeout -k 10 300 python bench.py --m $mm --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/bench_m${mm}_s${mode}.err | tail -1 > $OUT/bench_m${mm}_s${mode}.json
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_m${mm}_s${mode}.json"))
    print("m=$mm sharded=$mode: %.4f ms/step, %.1f it/s, gemm %.1f us" % (d["ms_per_step"], d["value"], 1e3 * d["roofline"]["avg_launch_ms"]), {k: v for k, v in d.get("distributed", {}).items() if "ms" in k})
except Exception as e:
    print("m=$mm sharded=$mode FAILED", e)
PY
  done
done
BMF_LIB=libbmf_stamp.so timeout -k 10 200 python scripts/gemm_i8_microbench.py 30 2>&1 | tee $OUT/stamps.txt
BMF_LIB=libbmf_stamp.so SHAPE=12500,20000,64 timeout -k 10 200 python scripts/gemm_i8_microbench.py 30 2>&1 | tee $OUT/stamps_12500.txt
