#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s6; mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest.log
grep -E "passed|failed|^FAILED|differs|Error|c3 parity" $OUT/pytest.log | tail -15
for cfg in "0" "1"; do
  echo "== bench fused $cfg"
  BMF_I8_FUSED_PLANES=$cfg timeout -k 10 300 python bench.py --steps 30 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 2>$OUT/bench_$cfg.err | tail -1 > $OUT/bench_$cfg.json
  python -c "
import json; d=json.load(open('$OUT/bench_$cfg.json')); print('%.4f ms/step %.1f it/s gemm %.1f us cold %.1f' % (d['ms_per_step'], d['value'], 1e3*d['roofline']['avg_launch_ms'], d['cold_start']['value']))"
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
