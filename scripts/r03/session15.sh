#!/bin/bash
# kernel statistics of the loop at the shard size of 8 GPUs (12 500 rows) and a timeline of one iteration
set -o pipefail
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/s15; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python3 $GRAFT_REPO_ROOT/bench.py --m 12500 --steps 60 --warmup 5 --cpu-rows 0 --traffic 0 --secondary 0 --sustained 0 --alt-operands none > $OUT/p.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/p/*/*kernel_stats.csv")[0]
for row in csv.DictReader(open(f)):
    n = row["Name"]
    if "anonymous" in n and "at::" not in n:
        print("%-62s calls %5s avg %8.1f us min %8.1f max %8.1f" % (n.split("(anonymous namespace)::")[1][:60], row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
t = glob.glob("$OUT/p/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
# one iteration late in the run: from a finalize to the next
idx = [i for i, r in enumerate(rows) if "finalize_kernel" in r["Kernel_Name"]]
a, b = idx[-12], idx[-11]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(anonymous namespace)::")[-1][:40]
    print("  +%7.1f us  gap %5.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n))
    prev_end = e
PY
