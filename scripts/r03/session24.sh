#!/bin/bash
# kernel statistics with the Gram tiles out of the epilogue (fused) vs not
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s24; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for f in 0 1; do
BMF_I8_FUSED_GRAM=$f rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 --alt-operands none > $OUT/p$f.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/p$f/*/*kernel_stats.csv")[0]
print("== fused_gram $f")
for row in csv.DictReader(open(f)):
    n = row["Name"]
    if "anonymous" in n and "at::" not in n and int(row["Calls"]) > 100:
        print("%-62s calls %5s avg %8.1f us" % (n.split("(anonymous namespace)::")[1][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done
