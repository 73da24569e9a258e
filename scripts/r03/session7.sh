#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s7; mkdir -p $OUT
export TMPDIR=/tmp
python scripts/r03/epi_bench.py 2>&1 | tail -1 | tee $OUT/epi.txt
python -m pytest tests/test_kernels_gpu.py tests/test_penalty_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
grep -E "passed|failed|^FAILED" $OUT/pytest.log | tail -5
for sh in 0.5 0.58 0.63 0.68; do
  echo "== cover share $sh"; BMF_COVER_OLD_SHARE=$sh python scripts/cover_microbench.py 2>&1 | grep "popc= [0248]" | tr '\n' ' '; echo
done | tee $OUT/cover_share.txt
for sh in 0.5 0.63; do
  echo "== bench cover share $sh"
  BMF_COVER_OLD_SHARE=$sh timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/bench_$sh.err | tail -1 > $OUT/bench_$sh.json
  python -c "
import json; d=json.load(open('$OUT/bench_$sh.json')); print('%.4f ms/step %.1f it/s gemm %.1f us cold %.1f' % (d['ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms'], d['cold_start']['value']))"
done | tee $OUT/bench_ab.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 --preheat 0 > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof/*/*kernel_stats.csv")[0]
for row in csv.DictReader(open(f)):
    n = row["Name"]
    if "anonymous" in n and "at::" not in n:
        print("%-62s calls %5s avg %8.1f us min %8.1f max %8.1f" % (n.split("(anonymous namespace)::")[1][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
PY
