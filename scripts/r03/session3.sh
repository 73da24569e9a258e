#!/bin/bash
# Round 3, GPU session 3: uneven stream-K slices for the two workgroups of a CU (int8 GEMM): correctness, then the share swept
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s3; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py tests/test_penalty_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
for a in 0.5 0.58 0.61 0.63 0.65 0.68; do
  echo "== share $a"
  BMF_I8_OLD_SHARE=$a timeout -k 10 200 python scripts/gemm_i8_microbench.py 40 2>&1 | tail -1
done | tee $OUT/share_sweep.txt
BMF_I8_OLD_SHARE=0.63 BMF_LIB=libbmf_stamp.so timeout -k 10 200 python scripts/gemm_i8_microbench.py 20 2>&1 | grep -v "^   xcc" | tee $OUT/stamps_063.txt
for a in 0.5 0.63; do
  echo "== bench share $a"
  BMF_I8_OLD_SHARE=$a timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/bench_$a.err | tail -1 > $OUT/bench_$a.json
  python -c "
import json; d=json.load(open('$OUT/bench_$a.json')); print('%.4f ms/step %.1f it/s gemm %.1f us' % (d['ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms']), d['checks'])"
done | tee $OUT/bench_share.txt
