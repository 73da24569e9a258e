#!/bin/bash
# masked update at MovieLens-1M shape: kernel statistics
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s26; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/scripts/masked_bench.py > $OUT/p.log 2>&1
grep "iteration" $OUT/p.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/p/*/*kernel_stats.csv")[0]
for row in csv.DictReader(open(f)):
    n = row["Name"]
    if int(row["Calls"]) >= 50:
        short = n.split("(anonymous namespace)::")[-1][:68]
        print("%-70s calls %4s avg %9.1f us total %8.2f ms" % (short, row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6))
PY
